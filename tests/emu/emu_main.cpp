// TEST INFRASTRUCTURE ONLY: runs rapid_amd/csrc/tally_kernel.h under the SIMT emulator (tests/emu/hip/).
#include <hip/hip_runtime.h>  // resolves to tests/emu/hip/hip_runtime.h

#include "tally_kernel.h"

namespace emu {
Wave* g_wave = nullptr;
Dim g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

static void trampoline() {
    Wave* w = g_wave;
    const int l = w->cur;
    w->body();
    w->done[l] = true;
    w->op[l] = OP_NONE;
    swapcontext(&w->ctx[l], &w->sched);
}

void run_block(unsigned block, unsigned grid, const std::function<void()>& body, uint64_t seed) {
    Wave w;
    g_wave = &w;
    w.body = body;
    w.rng = seed * 0x9E3779B97F4A7C15ull + 12345;
    g_blockIdx.x = block;
    g_gridDim.x = grid;
    g_blockDim.x = Wave::W;
    for (int l = 0; l < Wave::W; ++l) {
        w.stacks[l].resize(256 * 1024);
        w.done[l] = false;
        w.op[l] = OP_NONE;
        getcontext(&w.ctx[l]);
        w.ctx[l].uc_stack.ss_sp = w.stacks[l].data();
        w.ctx[l].uc_stack.ss_size = w.stacks[l].size();
        w.ctx[l].uc_link = &w.sched;
        makecontext(&w.ctx[l], trampoline, 0);
    }
    int order[Wave::W];
    for (int l = 0; l < Wave::W; ++l) order[l] = l;
    for (;;) {
        // random lane order for this phase
        for (int i = Wave::W - 1; i > 0; --i) {
            w.rng = w.rng * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(order[i], order[(w.rng >> 33) % (uint64_t)(i + 1)]);
        }
        bool any = false;
        for (int i = 0; i < Wave::W; ++i) {
            const int l = order[i];
            if (w.done[l]) continue;
            any = true;
            w.cur = l;
            g_threadIdx.x = (unsigned)l;
            swapcontext(&w.sched, &w.ctx[l]);
        }
        if (!any) break;
        // every live lane is now parked at a collective (or finished): they must all be at the same one
        int op = OP_NONE;
        uint64_t ballot = 0;
        for (int l = 0; l < Wave::W; ++l) {
            if (w.done[l]) continue;
            if (op == OP_NONE) op = w.op[l];
            if (w.op[l] != op) {
                std::fprintf(stderr, "emu: divergent wave op (lane %d at %d, expected %d)\n", l, w.op[l], op);
                std::abort();
            }
            if (op == OP_BALLOT && w.arg[l]) ballot |= 1ull << l;
        }
        for (int l = 0; l < Wave::W; ++l) {
            if (w.done[l]) continue;
            if (op == OP_BALLOT) w.res[l] = ballot;
            else if (op == OP_SHFL) w.res[l] = w.done[w.aux[l]] ? 0 : w.arg[w.aux[l]];
            else w.res[l] = 0;
        }
    }
    g_wave = nullptr;
}
}  // namespace emu

// dynamic LDS segment of the (single) resident block (`extern __shared__ smem[]` inside namespace rapid)
namespace rapid {
__attribute__((aligned(16))) unsigned char smem[160 * 1024];
}
using rapid::smem;

extern "C" {
int emu_tally_lds_bytes(int n_nodes) { return rapid::tally_lds_bytes(n_nodes); }

// Runs the population kernel for receivers [r0, r1) one block at a time.
int emu_tally_run(const unsigned char* records, unsigned long long records_bytes, const long long* rec_off,
                  int n_receivers, int n_nodes, int K, int H, int L, long long cfg_id,
                  const unsigned short* state_template, const int* obs, const int* subj, int* emit_batch,
                  int* num_proposals, int* prop_count, unsigned long long* fingerprint, int* props, int prop_cap,
                  unsigned long long* stats, int force_exact, unsigned long long seed) {
    if (rapid::tally_lds_bytes(n_nodes) > (int)sizeof(smem)) return -5;
    rapid::TallyParams p;
    p.records = records;
    p.records_bytes = records_bytes;
    p.rec_off = rec_off;
    p.n_receivers = n_receivers;
    p.n_nodes = n_nodes;
    p.K = K;
    p.H = H;
    p.L = L;
    p.cfg_id = cfg_id;
    p.state_template = state_template;
    p.obs = obs;
    p.subj = subj;
    p.emit_batch = emit_batch;
    p.num_proposals = num_proposals;
    p.prop_count = prop_count;
    p.fingerprint = fingerprint;
    p.props = props;
    p.prop_cap = prop_cap;
    p.stats = stats;
    p.force_exact = force_exact;
    for (int r = 0; r < n_receivers; ++r) {
        std::memset(smem, 0xCD, sizeof(smem));  // poison: the kernel must initialise what it reads
        emu::run_block((unsigned)r, (unsigned)n_receivers, [&] { rapid::tally_population_kernel(p); }, seed + (unsigned)r);
    }
    return 0;
}

int emu_cd_run(unsigned short* state, int* scal, const unsigned char* alerts, int n_alerts, int n_nodes, int K, int H,
               int L, const int* obs, const int* subj, int* out_idx, int out_cap, int* out_counts, int* out_n, int mode,
               unsigned long long seed) {
    rapid::CdParams p;
    p.state = state;
    p.scal = scal;
    p.alerts = alerts;
    p.n_alerts = n_alerts;
    p.n_nodes = n_nodes;
    p.K = K;
    p.H = H;
    p.L = L;
    p.obs = obs;
    p.subj = subj;
    p.out_idx = out_idx;
    p.out_cap = out_cap;
    p.out_counts = out_counts;
    p.out_n = out_n;
    p.mode = mode;
    emu::run_block(0, 1, [&] { rapid::cd_instance_kernel(p); }, seed);
    return 0;
}
}
