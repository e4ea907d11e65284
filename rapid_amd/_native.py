"""Loader / builder of librapid_mi355x.so (the C ABI of include/rapid_mi355x.h).

The product has no CPU fallback: if the shared library is missing, or the host has no gfx950 device, the
classes in rapid_amd.engine raise -- they never route to the oracle.
"""
import ctypes as C
import json
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAPID_MI355X_LIB") or os.path.join(_HERE, "librapid_mi355x.so")  # override: profiling builds
# the TEST build: the same sources with -DRAPID_TEST_BUILD -- plus the rapid_debug_* entry points, the probe kernels and the
# environment knobs of the measurement scripts.  Only tests/ and scripts/ load it (use_test_build()).
TEST_LIB_PATH = os.path.join(_HERE, "librapid_mi355x_test.so")
SRC_DIR = os.path.join(_HERE, "csrc")
SOURCES = ["engine.hip", "host_abi.cpp", "tally_kernel.h", "tally_probes.inc", "stream_probes.inc", "stream_load.h", "lds_dma.h", "index_kernels.h", "view_kernels.h", "vote_kernels.h", "wire.h", "consensus.h"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "rapid_mi355x.h")

OK, EINVAL, ENODE_EXISTS, ENODE_MISSING, EUUID_SEEN, ECAPACITY, EDEVICE, ESTATE, ECOLLISION = 0, -1, -2, -3, -4, -5, -6, -7, -8


class RapidError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("rapid_mi355x error %d: %s" % (code, msg))
        self.code = code


class NodeAlreadyInRingException(RapidError):  # R/MembershipView.java:502-506
    pass


class NodeNotInRingException(RapidError):  # R/MembershipView.java:508-512
    pass


class UUIDAlreadySeenException(RapidError):  # R/MembershipView.java:514-519
    pass


class IllegalArgumentException(RapidError, ValueError):  # R/MultiNodeCutDetector.java:52-55
    pass


_EXC = {EINVAL: IllegalArgumentException, ENODE_EXISTS: NodeAlreadyInRingException,
        ENODE_MISSING: NodeNotInRingException, EUUID_SEEN: UUIDAlreadySeenException}


def raise_for(code, msg=""):
    raise _EXC.get(code, RapidError)(code, msg)


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(path):
    if not os.path.exists(path):
        return True
    srcs = [os.path.join(SRC_DIR, s) for s in SOURCES] + [HEADER]
    if not all(os.path.exists(s) for s in srcs):
        return False  # sources absent (binary-only snapshot): use what is there
    return os.path.getmtime(path) < max(os.path.getmtime(s) for s in srcs)


def needs_build():
    return _stale(LIB_PATH) or _stale(TEST_LIB_PATH)


RESOURCES_PATH = os.path.join(os.path.dirname(LIB_PATH), "librapid_mi355x.resources.json")


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 cross-compiles without a GPU (about a minute): the product library and, in parallel, the test
    build.  The compiler's per-kernel resource report of the PRODUCT (VGPRs, scratch bytes per lane, LDS) is kept next to the
    library: tests/test_build.py reads it, because a tally kernel that starts spilling is a silent 20 % regression, not a
    build failure."""
    if not force and not needs_build():
        return LIB_PATH
    jobs = []
    for out, defs in ((LIB_PATH, []), (TEST_LIB_PATH, ["-DRAPID_TEST_BUILD"])):
        tmp = "%s.tmp%d" % (out, os.getpid())  # concurrent builders (pytest-xdist workers) must not share the output file
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I" + SRC_DIR] + defs + \
              ["-Rpass-analysis=kernel-resource-usage",
               os.path.join(SRC_DIR, "engine.hip"), os.path.join(SRC_DIR, "host_abi.cpp"), "-o", tmp, "-lrccl"]
        if verbose:
            print(" ".join(cmd))
        jobs.append((out, tmp, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for out, tmp, cmd, proc in jobs:
        _, err = proc.communicate()
        if proc.returncode != 0:
            sys.stderr.write(err)
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        if out == LIB_PATH:
            with open(tmp + ".json", "w") as f:
                json.dump(parse_resource_remarks(err), f, indent=1, sort_keys=True)
            os.replace(tmp + ".json", RESOURCES_PATH)
        os.replace(tmp, out)
    return LIB_PATH


def parse_resource_remarks(text):
    """-Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {"VGPRs": n, "ScratchSize [bytes/lane]": n, ...}}."""
    out, cur = {}, None
    for line in text.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].split("[-Rpass-analysis", 1)[0].strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            try:
                cur[k.strip()] = int(v.strip())
            except ValueError:
                cur[k.strip()] = v.strip()
    return out


_lib = None


class RoundResult(C.Structure):
    _fields_ = [("decided", C.c_int32), ("cut_size", C.c_int32), ("quorum", C.c_int32), ("membership_size", C.c_int32),
                ("votes_total", C.c_int64), ("votes_winner", C.c_int64), ("distinct_local", C.c_int32),
                ("reserved", C.c_int32), ("config_id", C.c_int64)]


class Rank(C.Structure):  # rapid_rank
    _fields_ = [("round", C.c_int32), ("node_index", C.c_int32)]


class ConsensusMsg(C.Structure):  # rapid_consensus_msg
    _fields_ = [("kind", C.c_int32), ("sender", C.c_int32), ("config_id", C.c_int64), ("rnd", Rank), ("vrnd", Rank),
                ("n_endpoints", C.c_int32), ("dest", C.c_int32)]


class ClassicRoundResult(C.Structure):  # rapid_classic_round_result
    _fields_ = [("decided", C.c_int32), ("chosen_acceptor", C.c_int32), ("promises_used", C.c_int32), ("rule", C.c_int32),
                ("messages", C.c_int64)]


class ClassicStart(C.Structure):  # rapid_classic_start
    _fields_ = [("step", C.c_int32), ("acceptor", C.c_int32), ("round", C.c_int32)]


class ClassicRoundsResult(C.Structure):  # rapid_classic_rounds_result
    _fields_ = [("decided_nodes", C.c_int32), ("agreed", C.c_int32), ("chosen_acceptor", C.c_int32), ("lost", C.c_int32),
                ("steps", C.c_int64), ("undelivered", C.c_int64), ("sent", C.c_int64 * 4), ("delivered", C.c_int64 * 4)]


class EngineConfig(C.Structure):
    _fields_ = [("n_max", C.c_int32), ("K", C.c_int32), ("H", C.c_int32), ("L", C.c_int32), ("device_id", C.c_int32),
                ("max_cut", C.c_int32)]


# every entry point declared in include/rapid_mi355x.h: name -> (restype, argtypes)
def _signatures():
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    p = C.c_void_p  # all array arguments are passed as raw addresses
    pi32, pi64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    return {
        "rapid_engine_create": (i32, [C.POINTER(EngineConfig), C.POINTER(vp)]),
        "rapid_engine_destroy": (None, [vp]),
        "rapid_last_error": (C.c_char_p, [vp]),
        "rapid_device_count": (i32, []),
        "rapid_engine_self_test": (i32, [vp]),
        "rapid_engine_comm_info": (i32, [vp, pi32, pi32]),
        "rapid_sim_new_round": (i32, [vp]),
        "rapid_sim_trust_alert_copies": (i32, [vp, i32]),
        "rapid_view_build": (i32, [vp, p, p, p, p, p, i32, p, i32, p, p, i32]),
        "rapid_view_register_endpoints": (i32, [vp, p, p, p, p, p, i32, pi32]),
        "rapid_view_is_safe_to_join": (i32, [vp, i32, i64, i64, pi32]),
        "rapid_view_ring_add": (i32, [vp, i32, i64, i64]),
        "rapid_view_ring_delete": (i32, [vp, i32]),
        "rapid_view_observers": (i32, [vp, i32, p, i32, pi32]),
        "rapid_view_subjects": (i32, [vp, i32, p, i32, pi32]),
        "rapid_view_expected_observers": (i32, [vp, i32, p, i32, pi32]),
        "rapid_view_ring_numbers": (i32, [vp, i32, i32, p, i32, pi32]),
        "rapid_view_ring": (i32, [vp, i32, p, i32, pi32]),
        "rapid_view_ring_key": (i32, [vp, i32, i32, pi64]),
        "rapid_view_is_host_present": (i32, [vp, i32, pi32]),
        "rapid_view_size": (i32, [vp, pi32]),
        "rapid_view_config_id": (i32, [vp, pi64]),
        "rapid_view_tables": (i32, [vp, p, p, p, i32]),
        "rapid_view_q4_at_risk": (i32, [vp, p, i32, p, i32, pi32]),
        "rapid_view_q4_emulation": (i32, [vp, i32]),
        "rapid_cd_create": (i32, [vp, i32, i32, i32, C.POINTER(vp)]),
        "rapid_cd_destroy": (None, [vp]),
        "rapid_cd_aggregate": (i32, [vp, p, i32, p, i32, p, pi32]),
        "rapid_cd_invalidate": (i32, [vp, p, i32, pi32]),
        "rapid_cd_num_proposals": (i32, [vp, pi32]),
        "rapid_cd_clear": (i32, [vp]),
        "rapid_sim_load_streams": (i32, [vp, p, p, i32]),
        "rapid_sim_load_streams_device": (i32, [vp, p, u64, p, i32]),
        "rapid_sim_set_alert_set": (i32, [vp, p, i64]),
        "rapid_sim_set_alert_set_device": (i32, [vp, p, u64, i64]),
        "rapid_sim_attach_streams_device": (i32, [vp, p, u64, p, i32]),
        "rapid_sim_generate": (i32, [vp, p, p, i32, p, p, i32, u64, i32]),
        "rapid_sim_round_tiled": (i32, [vp, p, p, i32, p, p, i32, i32, u64, i32, C.POINTER(RoundResult)]),
        "rapid_sim_round_tiled_info": (i32, [vp, p]),
        "rapid_sim_tally": (i32, [vp]),
        "rapid_sim_results": (i32, [vp, p, p, p, p, i32]),
        "rapid_sim_proposal": (i32, [vp, i32, p, i32, pi32]),
        "rapid_sim_count_votes": (i32, [vp, C.POINTER(RoundResult)]),
        "rapid_sim_decided_cut": (i32, [vp, p, i32, pi32]),
        "rapid_sim_round": (i32, [vp, i32, C.POINTER(RoundResult), pi64]),
        "rapid_sim_round_device": (i32, [vp, p, u64, p, i32, p, u64, i64, i32, i32, C.POINTER(RoundResult), pi64]),
        "rapid_apply_cut": (i32, [vp, p, i32, pi64]),
        "rapid_fast_round_create": (i32, [i64, i32, C.POINTER(vp)]),
        "rapid_fast_round_destroy": (None, [vp]),
        "rapid_fast_round_vote": (i32, [vp, i32, i64, p, i32, pi32]),
        "rapid_fast_round_decision": (i32, [vp, p, i32, pi32]),
        "rapid_comm_unique_id": (i32, [p]),
        "rapid_endpoint_map_create": (i32, [p, p, p, i32, C.POINTER(vp)]),
        "rapid_endpoint_map_destroy": (None, [vp]),
        "rapid_endpoint_map_lookup": (i32, [vp, p, i32, i32, pi32]),
        "rapid_decode_request": (i32, [p, i64, pi32, pi64, pi64]),
        "rapid_decode_batched_alerts": (i32, [vp, p, i64, i32, p, p, p, i32, pi32, pi32]),
        "rapid_decode_batched_alerts_ex": (i32, [vp, p, i64, i32, p, p, p, p, p, i32, pi32, pi32]),
        "rapid_endpoint_map_add": (i32, [vp, p, i32, i32, pi32]),
        "rapid_endpoint_map_add_wire": (i32, [vp, p, i64, pi32]),
        "rapid_endpoint_map_size": (i32, [vp]),
        "rapid_endpoint_map_get": (i32, [vp, i32, p, i32, pi32, pi32]),
        "rapid_decode_fast_round_vote": (i32, [vp, p, i64, pi32, pi64, p, i32, pi32]),
        "rapid_consensus_create": (i32, [i32, i32, i64, i32, C.POINTER(vp)]),
        "rapid_consensus_destroy": (None, [vp]),
        "rapid_consensus_propose": (i32, [vp, p, i32]),
        "rapid_consensus_handle": (i32, [vp, C.POINTER(ConsensusMsg), p]),
        "rapid_consensus_start_classic_round": (i32, [vp]),
        "rapid_consensus_start_phase1a": (i32, [vp, i32]),
        "rapid_consensus_poll": (i32, [vp, C.POINTER(ConsensusMsg), p, i32, pi32]),
        "rapid_consensus_decision": (i32, [vp, p, i32, pi32]),
        "rapid_consensus_fallback_delay_ms": (i32, [i32, i64, C.c_double, pi64]),
        "rapid_paxos_select_proposal": (i32, [i32, p, p, p, i32, pi32]),
        "rapid_classic_round_population": (i32, [i32, i32, p, p, p, C.POINTER(ClassicRoundResult)]),
        "rapid_classic_rounds_population": (i32, [i32, i32, p, p, p, p, i32, p, p, i64, u64, C.c_double, C.POINTER(ClassicRoundsResult), p]),
        "rapid_decode_consensus_message": (i32, [vp, i32, p, i64, C.POINTER(ConsensusMsg), p, i32]),
        "rapid_encode_consensus_request": (i32, [vp, C.POINTER(ConsensusMsg), p, p, i64, pi64]),
        "rapid_engine_comm_init": (i32, [vp, p, i32, i32]),
        "rapid_engine_stream": (vp, [vp]),
        "rapid_engine_sync": (i32, [vp]),
        "rapid_sim_stats": (i32, [vp, p]),
        "rapid_sim_time_tally": (i32, [vp, i32, C.POINTER(C.c_float)]),
        "rapid_sim_set_force_exact": (i32, [vp, i32]),
        "rapid_sim_index_info": (i32, [vp, p, C.POINTER(C.c_float)]),
        "rapid_sim_pass_times": (i32, [vp, p]),
    }


def _test_signatures():
    """the entry points only the test build exports (include/rapid_mi355x.h, section RAPID_TEST_BUILD)"""
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    p, pi32, pi64 = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    return {
        "rapid_debug_block_stats": (i32, [vp, p, i32, pi32]),
        "rapid_debug_vote_segment": (i32, [vp, p, i64, pi64]),
        "rapid_debug_vote_merge": (i32, [vp, p, i32, p, pi32]),
        "rapid_debug_read_records": (i32, [vp, i64, i32, p, p]),
        "rapid_debug_stream_probe": (i32, [vp, i32, i32, i32, C.POINTER(C.c_float)]),
        "rapid_debug_device_fault": (i32, [vp]),
    }


SIGNATURES = _signatures()
TEST_SIGNATURES = _test_signatures()


def _load(path, with_test_symbols):
    L = C.CDLL(path)
    sigs = dict(SIGNATURES, **TEST_SIGNATURES) if with_test_symbols else SIGNATURES
    for name, (res, args) in sigs.items():
        f = getattr(L, name)  # AttributeError here == the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    return L


def lib():
    """Loads the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RapidError(EDEVICE, "librapid_mi355x.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # (a profiling / variant build is usually a test build: it carries the debug entry points, too)
    if os.path.basename(LIB_PATH) != "librapid_mi355x.so":
        try:
            _lib = _load(LIB_PATH, True)
            return _lib
        except AttributeError:
            pass
    _lib = _load(LIB_PATH, False)
    return _lib


def use_test_build():
    """Engines created from now on come from librapid_mi355x_test.so (tests/ and scripts/ that need rapid_debug_* or the probes).
    Returns the previous library path; use_library(path) switches back."""
    if os.environ.get("RAPID_MI355X_LIB"):  # a profiling build chosen by the environment is built with -DRAPID_TEST_BUILD already
        return LIB_PATH
    return use_library(TEST_LIB_PATH)


def use_library(path):
    global _lib, LIB_PATH
    prev = LIB_PATH
    if path != LIB_PATH:
        if not os.path.exists(path):
            raise RapidError(EDEVICE, "%s is not built" % path)
        LIB_PATH = path
        _lib = None
    return prev
