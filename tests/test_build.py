"""The compiler's verdict on the hot kernels, checked on the CPU: rapid_amd._native.build() keeps hipcc's per-kernel resource
report next to the library, and the tally kernel must not spill.  Its eight instantiations sit between 86 and 133 VGPRs
under budgets of 128 (16 waves per CU) and 168 (12), so an innocent-looking edit can push one over: the reloads then land inside the
receiver loop, count against vmcnt like stream loads, and cost ~20 % (measured in round 2: 0.42 -> 0.52 ms on C3b for the
per-delivery-filter kernels) without any test failing."""
import json
import os

from rapid_amd import _native as N


def resources():
    if not os.path.exists(N.RESOURCES_PATH) or N.needs_build():
        N.build(force=True)
    with open(N.RESOURCES_PATH) as f:
        return json.load(f)


def test_tally_kernels_fit_the_register_file_without_scratch():
    res = resources()
    tally = {k: v for k, v in res.items() if "tally_population_kernel" in k}
    # {20-byte boundary records looked up in memory / direct tables / compressed tables, resolved 8-byte records of the generator}
    # x {filter per delivery, trusted copies} + {in memory, resolved} x {filter, trusted} with two slots per LDS word
    # ... + the hashed dictionary in LDS (mode 4) x {filter, trusted}, packed
    # ... + trusted boundary records known to carry the engine's configuration id (kCurrent): memory / direct / compressed, memory packed
    assert len(tally) == 18, sorted(tally)
    for name, r in tally.items():
        assert r["ScratchSize [bytes/lane]"] == 0, (name, r)
        # 16 waves per CU = 4 per SIMD = 128 VGPRs; the per-delivery filter over compressed tables on boundary records is
        # launched with at most 12 (tally_kernel.h: tally_max_waves): 3 per SIMD = 168; the packed instantiations (a handful of
        # receivers per CU whatever the registers: their LDS decides) with at most 8: 2 per SIMD = 256
        twelve = "tally_population_kernelILi2ELb0ELi1E" in name
        packed = name.split("tally_population_kernelILi")[1][12] == "1"
        assert r["VGPRs"] <= (256 if packed else 168 if twelve else 128), (name, r)
        assert r["Occupancy [waves/SIMD]"] >= (2 if packed else 3 if twelve else 4), (name, r)


def test_no_kernel_of_the_product_uses_scratch_memory():
    """EVERY kernel in librapid_mi355x.so, the library sort's included (round 6: the default configuration of rocPRIM's segmented
    radix sort spilled 148 bytes per lane on gfx950 -- the only kernels of the product that made the runtime set up a scratch arena
    on the queue, inside the first call every user makes; csrc/engine.hip: RingSortConfig)."""
    res = resources()
    ours = {k: v for k, v in res.items() if k.startswith("_ZN5rapid")}
    assert len(ours) >= 20 and len(res) > len(ours)  # (the report covers the library's kernels, too)
    spilling = {k[:160]: v["ScratchSize [bytes/lane]"] for k, v in res.items() if v.get("ScratchSize [bytes/lane]", 0) != 0}
    assert not spilling, spilling


# (dictionary mode, trusted, record format, packed): 0 / 1 / 2 = tables in memory / direct in LDS / compressed in LDS over 20-byte
# boundary records (format 1); 3 = resolved 8-byte records (format 0); packed = two slots per LDS word
# 4 = hashed buckets in LDS (packed rounds, opt-in).  The packed instantiations keep four windows of the stream in flight (round 5).
# fifth parameter (kCurrent): trusted boundary records whose configuration ids are never loaded -- 16 registers fewer
EXPECTED_VGPRS = {(0, False, 1, False): 126, (0, False, 1, True): 246, (0, True, 1, False): 107, (0, True, 1, True): 231, (1, False, 1,
                  False): 122, (1, True, 1, False): 106, (2, False, 1, False): 137, (2, True, 1, False): 117, (3, False, 0, False): 94, (3,
                  False, 0, True): 153, (3, True, 0, False): 90, (3, True, 0, True): 146, (4, False, 1, True): 256, (4, True, 1, True): 242,
                  (0, True, 1, False, True): 100, (0, True, 1, True, True): 183, (1, True, 1, False, True): 99, (2, True, 1, False, True):
                  110}


def test_register_allocation_of_the_tally_kernel_is_the_measured_one():
    """A canary, not a law: changes that leave every result identical can still cost 20-30 % -- in round 2 once through
    spills after an edit of the table-staging loop, once because a rewrite of that loop left a table pointer aimed at global
    memory instead of its LDS copy (89 -> 93 VGPRs, 0.214 -> 0.285 ms on C3b) -- while every parity test stayed green.
    These are the counts of the build whose timings are in profiles/r06_*; if they move, time the tally kernel on a GPU
    (scripts/ab_variants.py prints it in seconds) before accepting the new numbers here."""
    res = resources()
    got = {}
    for name, r in res.items():
        if "tally_population_kernel" in name:
            t = name.split("tally_population_kernelILi")[1]  # <dictionary mode>ELb<trusted>ELi<record format>ELb<packed>EEEv...
            key = (int(t[0]), t[4] == "1", int(t[8]), t[12] == "1")
            got[key + (True,) if t[16] == "1" else key] = r["VGPRs"]
    assert got == EXPECTED_VGPRS, got


def test_product_is_compiled_without_measurement_hooks():
    """The hot kernel's source holds hook POINTS and their empty defaults; the bodies (phase timers, time stamps, trace lines) live in
    csrc/tally_probes.inc, which only -DRAPID_MEASUREMENT_BUILD pulls in -- and the product refuses that define: one stray -D must not
    change what librapid_mi355x.so computes (several of round 4's probes voided the results)."""
    import re
    import subprocess
    src = open(os.path.join(N.SRC_DIR, "tally_kernel.h")).read()
    assert "RAPID_PROBE_" not in src and '#include "tally_probes.inc"' in src
    body = src.split('#include "tally_probes.inc"', 1)[1]
    for word in ("RAPID_PHASE_TIMERS", "RAPID_BLOCK_STAMPS", "RAPID_TRACE\b", "fprintf"):
        assert not re.search(word, body), word
    # the product's command line (rapid_amd/_native.py: build) carries no -D at all; the test build exactly one
    native = open(os.path.join(os.path.dirname(N.SRC_DIR), "_native.py")).read()
    assert re.findall(r'"-D[A-Z_]+"', native) == ['"-DRAPID_TEST_BUILD"']
    # ... and a product translation unit that is handed a hook switch anyway does not compile
    for define in ("-DRAPID_MEASUREMENT_BUILD", "-DRAPID_PHASE_TIMERS", "-DRAPID_TRACE"):
        r = subprocess.run([N.hipcc(), "--offload-arch=gfx950", "-std=c++17", "-E", "--cuda-host-only", define, "-I" + N.SRC_DIR,
                            os.path.join(N.SRC_DIR, "engine.hip"), "-o", os.devnull], capture_output=True, text=True)
        assert r.returncode != 0 and "measurement hooks" in r.stderr, (define, r.stderr[-300:])
