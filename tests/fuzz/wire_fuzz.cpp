// TEST INFRASTRUCTURE ONLY: mutation fuzzer for the wire decoders of rapid_amd/csrc/host_abi.cpp (the code that reads
// bytes from the network).  Built with -fsanitize=address,undefined by tests/test_wire_fuzz.py and fed a corpus of
// valid serialized messages; every decoder must return an error code or a result on any input -- never read out of
// bounds, overflow or crash.
//   wire_fuzz <corpus file> <iterations> <seed>
// corpus file: repeated { u32 kind; u32 length; bytes } with kind = 0 for a whole RapidRequest, else the content case.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rapid_mi355x.h"

struct Rng {
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return n ? (uint32_t)(next() % n) : 0u; }
};

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<std::pair<uint32_t, std::vector<uint8_t>>> corpus;
    for (;;) {
        uint32_t head[2];
        if (std::fread(head, 4, 2, f) != 2) break;
        std::vector<uint8_t> b(head[1]);
        if (head[1] && std::fread(b.data(), 1, head[1], f) != head[1]) return 2;
        corpus.emplace_back(head[0], std::move(b));
    }
    std::fclose(f);
    if (corpus.empty()) return 2;
    const long iterations = std::atol(argv[2]);
    Rng rng{(uint64_t)std::atoll(argv[3])};

    // the endpoint map of the corpus: hosts "h<i>" port 5000 + i, i < 64
    std::string blob;
    std::vector<int32_t> off{0}, ports;
    for (int i = 0; i < 64; ++i) {
        blob += "h" + std::to_string(i);
        off.push_back((int32_t)blob.size());
        ports.push_back(5000 + i);
    }
    rapid_endpoint_map* map = nullptr;
    if (rapid_endpoint_map_create(reinterpret_cast<const uint8_t*>(blob.data()), off.data(), ports.data(), 64, &map) != RAPID_OK) return 3;

    long ok = 0, rejected = 0, polled = 0;
    rapid_consensus* node = nullptr;  // a consensus instance that is handed whatever decodes (re-created now and then)
    std::vector<rapid_alert_record> recs(64);
    std::vector<int64_t> hi(64), lo(64);
    std::vector<int32_t> eps(64);
    for (long it = 0; it < iterations; ++it) {
        const auto& seed = corpus[rng.below((uint32_t)corpus.size())];
        std::vector<uint8_t> m = seed.second;
        const uint32_t n_mut = rng.below(4);  // 0 = the valid message itself
        for (uint32_t k = 0; k < n_mut; ++k) {
            switch (rng.below(5)) {
                case 0: if (!m.empty()) m[rng.below((uint32_t)m.size())] ^= (uint8_t)(1u << rng.below(8)); break;
                case 1: if (!m.empty()) m[rng.below((uint32_t)m.size())] = (uint8_t)rng.next(); break;
                case 2: m.resize(rng.below((uint32_t)m.size() + 1)); break;
                case 3: m.insert(m.begin() + rng.below((uint32_t)m.size() + 1), (uint8_t)(rng.below(2) ? 0xFF : rng.next())); break;
                default: if (m.size() > 1) m.erase(m.begin() + rng.below((uint32_t)m.size())); break;
            }
        }
        // hand the decoders an exactly-sized heap copy so that any overrun is caught
        uint8_t* buf = static_cast<uint8_t*>(std::malloc(m.size() ? m.size() : 1));
        std::memcpy(buf, m.data(), m.size());
        const int64_t len = (int64_t)m.size();
        int32_t kind = (int32_t)seed.first;
        int64_t poff = 0, plen = len;
        int rc = RAPID_OK;
        if (kind == 0) {
            rc = rapid_decode_request(buf, len, &kind, &poff, &plen);
            if (rc == RAPID_OK && (poff < 0 || plen < 0 || poff + plen > len)) return 4;  // payload outside the message
        }
        if (rc == RAPID_OK) {
            const uint8_t* p = buf + poff;
            int32_t n = 0, sender = 0;
            int64_t cfg = 0;
            if (kind == RAPID_MSG_BATCHED_ALERT) {
                rc = rapid_decode_batched_alerts(map, p, plen, 10, recs.data(), hi.data(), lo.data(), (int32_t)recs.size(), &n, &sender);
                if (rc == RAPID_OK) {
                    if (n < 0 || n > (int32_t)recs.size()) return 5;
                    for (int i = 0; i < n; ++i)
                        if (recs[i].src >= 64u || recs[i].dst >= 64u || recs[i].ring_mask >= (1u << 10) || recs[i].status > 1) return 6;
                }
            } else if (kind == RAPID_MSG_FAST_ROUND_2B && rng.below(2)) {
                rc = rapid_decode_fast_round_vote(map, p, plen, &sender, &cfg, eps.data(), (int32_t)eps.size(), &n);
                if (rc == RAPID_OK && (n < 0 || n > (int32_t)eps.size())) return 5;
            } else if (kind >= RAPID_MSG_FAST_ROUND_2B && kind <= RAPID_MSG_PHASE2B) {
                rapid_consensus_msg head;
                rc = rapid_decode_consensus_message(map, kind, p, plen, &head, eps.data(), (int32_t)eps.size());
                if (rc == RAPID_OK) {
                    if (head.n_endpoints < 0 || head.n_endpoints > (int32_t)eps.size() || head.sender < 0 || head.sender >= 64) return 5;
                    // what decodes must encode, and decode again to the same message
                    std::vector<uint8_t> out(4096);
                    int64_t olen = 0;
                    if (rapid_encode_consensus_request(map, &head, eps.data(), out.data(), (int64_t)out.size(), &olen) != RAPID_OK) return 7;
                    int32_t k2 = 0;
                    int64_t o2 = 0, l2 = 0;
                    if (rapid_decode_request(out.data(), olen, &k2, &o2, &l2) != RAPID_OK || k2 != kind) return 8;
                    rapid_consensus_msg again;
                    std::vector<int32_t> eps2(64);
                    if (rapid_decode_consensus_message(map, k2, out.data() + o2, l2, &again, eps2.data(), 64) != RAPID_OK) return 9;
                    if (std::memcmp(&again, &head, sizeof head) != 0 || !std::equal(eps2.begin(), eps2.begin() + head.n_endpoints, eps.begin())) return 10;
                    // ... and the state machine must digest it, whatever its ranks and values are
                    if (!node || rng.below(512) == 0) {
                        rapid_consensus_destroy(node);
                        if (rapid_consensus_create((int32_t)rng.below(64), (int32_t)rng.next(), head.config_id, 1 + (int32_t)rng.below(12), &node) != RAPID_OK) return 12;
                        if (rng.below(2) && rapid_consensus_propose(node, eps.data(), head.n_endpoints) != RAPID_OK) return 12;
                        if (rng.below(2) && rapid_consensus_start_classic_round(node) != RAPID_OK) return 12;
                    }
                    head.config_id = rng.below(8) ? head.config_id : head.config_id + 1;
                    if (rapid_consensus_handle(node, &head, eps.data()) != RAPID_OK) return 13;
                    for (;;) {
                        rapid_consensus_msg o;
                        int32_t got = 0;
                        const int prc = rapid_consensus_poll(node, &o, eps2.data(), 64, &got);
                        if (prc != RAPID_OK || !got) break;
                        if (o.kind < RAPID_MSG_FAST_ROUND_2B || o.kind > RAPID_MSG_PHASE2B || o.n_endpoints < 0 || o.n_endpoints > 64) return 14;
                        ++polled;
                    }
                    int32_t dn = 0;
                    const int drc = rapid_consensus_decision(node, eps2.data(), 64, &dn);
                    if (drc != RAPID_OK && drc != RAPID_ESTATE) return 15;
                }
            }
        }
        (rc == RAPID_OK ? ok : rejected)++;
        std::free(buf);
    }
    rapid_consensus_destroy(node);
    rapid_endpoint_map_destroy(map);
    std::printf("accepted %ld rejected %ld, consensus messages sent %ld\n", ok, rejected, polled);
    return ok > 0 && rejected > 0 && polled > 0 ? 0 : 11;
}
