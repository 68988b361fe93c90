"""Thin host wrappers: torch tensors (storage only) -> C-ABI calls of libdenet_hip.so.

Tensors are NHWC fp32, contiguous, on the current HIP device. No torch arithmetic happens here; every
function enqueues hand-written HIP kernels on the current torch stream and returns.
"""
import os

import torch

from . import lib as _lib
from .lib import check, ptr, stream_ptr


def _L():
    return _lib.load()


def empty(*shape, dtype=torch.float32):
    return torch.empty(*shape, dtype=dtype, device="cuda")


class Workspace:
    """Grow-only scratch buffers shared by all kernels of one stream (conv split-K slices, BN partials, ...)."""

    def __init__(self):
        self.bufs = {}

    def get(self, name, nbytes):
        nbytes = int(nbytes)
        buf = self.bufs.get(name)
        if buf is None or buf.numel() < nbytes:
            buf = torch.zeros(max(nbytes, 1), dtype=torch.uint8, device="cuda")
            self.bufs[name] = buf
        return buf


WS = Workspace()
WGRAD_WS_BYTES = 1024 << 20


def kernel_symbol(kind, a, b, c):
    """a profile record -> the kernel's name as rocprofv3 prints it (one row of profiles/*_kernel_stats.md per name)"""
    if kind == 10:
        return "wino2f_ws_kernel<%s>" % ("true" if a else "false")
    if kind == 14:
        return "%s<%d>" % ({32: "wino4f_kernel_32", 64: "wino4f_kernel_64", 33: "wino4f_kernel_32x2", 34: "wino4f_kernel_32k"}[a], b)
    if kind == 15:
        return "wino4g_kernel<%d>" % c
    if kind == 16:
        return "wino4t_kernel<%d>" % a
    if kind == 17:
        return "dgrad_s2_kernel<%d>" % a
    fixed = {11: "wino2f_wgrad_kernel", 12: "stem_fwd_kernel", 13: "stem_wgrad_kernel"}.get(kind)
    return fixed or "igemm_kernel<%d, %d, %d, 2, 2, %d>" % (kind, a, b, c)


class KernelProfile:
    """Live per-kernel timing (bench.py's roofline leg). The C side records one HIP event pair around every igemm
    kernel launch, on the stream it is launched on (denet_conv_profile, include/denet_hip.h); this side keeps the
    algorithmic FLOPs (2 * MACs of the layer) of the same launches, in the same order."""

    def __init__(self, alone=True):
        """alone (the roofline leg): while the profile is recorded every kernel runs alone on ONE stream (the second stream of the
        backward sweep is off), so that a launch's duration is the kernel's own. alone=False: the step keeps its two chains and
        the event pairs measure every matrix kernel INSIDE the step - beside the other chain's kernels, waits for CU slots included
        (bench.py: roofline.dominant_by_time_in_step)"""
        self.flops = []
        self.alone = bool(alone)
        check(_L().denet_conv_profile(1), "conv_profile")

    def add(self, flops):
        self.flops.append(flops)

    def summary(self):
        import ctypes
        torch.cuda.synchronize()
        L = _L()
        n = L.denet_conv_profile_count()
        assert n == len(self.flops), "profiled launches (%d) != convolution calls (%d)" % (n, len(self.flops))
        agg = {}
        ms, v = ctypes.c_float(), [ctypes.c_int() for _ in range(4)]
        for i, flops in enumerate(self.flops):
            check(L.denet_conv_profile_read(i, ctypes.byref(ms), *[ctypes.byref(x) for x in v]), "conv_profile_read")
            name = kernel_symbol(*[x.value for x in v])
            a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0})
            a["launches"] += 1
            a["ms"] += ms.value
            a["flops"] += flops
        check(L.denet_conv_profile(0), "conv_profile")
        return agg


PROFILE = None


class LaunchTrace:
    """Which matrix kernel every convolution pass launches, in launch order (denet_conv_profile(2): the C side notes the
    instantiation of each launch - no events, no change of streams or timing). mark(label) closes the stretch of launches
    since the previous mark; by_label() -> {label: [kernel symbols as rocprofv3 prints them]}. Used by
    denet_amd.model.audit (bench.py's kernels_used, the parity tests' per-layer assertions)."""

    def __init__(self):
        self.marks = []
        self.symbols = None

    def __enter__(self):
        assert PROFILE is None, "a live kernel profile is being recorded"
        check(_L().denet_conv_profile(2), "conv_profile")
        return self

    def mark(self, label):
        self.marks.append((label, int(_L().denet_conv_profile_count())))

    def __exit__(self, *a):
        import ctypes
        L = _L()
        n = int(L.denet_conv_profile_count())
        ms, v = ctypes.c_float(), [ctypes.c_int() for _ in range(4)]
        self.symbols = []
        for i in range(n):
            check(L.denet_conv_profile_read(i, ctypes.byref(ms), *[ctypes.byref(x) for x in v]), "conv_profile_read")
            self.symbols.append(kernel_symbol(*[x.value for x in v]))
        check(L.denet_conv_profile(0), "conv_profile")
        return False

    def by_label(self):
        out, lo = {}, 0
        for label, hi in self.marks:
            out.setdefault(label, []).extend(self.symbols[lo:hi])
            lo = hi
        if lo < len(self.symbols):
            out.setdefault(None, []).extend(self.symbols[lo:])
        return out


_COPY_STREAM = None


SIDE_STREAMS = []          # probed streams that run beside the compute stream (init_streams)


def side_stream(i=0):
    """a stream that really runs beside the compute stream (for small asynchronous copies)"""
    init_streams()
    return SIDE_STREAMS[i % len(SIDE_STREAMS)] if SIDE_STREAMS else torch.cuda.Stream()


def _pair_time(a, b, us):
    """wall time (us) of one idle kernel on stream a followed/accompanied by one on stream b (b ordered behind a's start)"""
    L = _L()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(a)
    check(L.denet_spin(us, a.cuda_stream), "spin")
    if b is not a:
        b.wait_event(e0)
    check(L.denet_spin(us, b.cuda_stream), "spin")
    if b is not a:
        e2 = torch.cuda.Event()
        e2.record(b)
        a.wait_event(e2)
    e1.record(a)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3


def _runs_beside(stream, main, us=300):
    """True if a kernel on `stream` really overlaps one on `main` (they sit on different hardware queues): the pair must take
    clearly less than the same two kernels back to back on `main` (best of three: other work on the device only ever makes
    a pair slower)"""
    if stream is main:
        return False
    torch.cuda.synchronize()
    serial = min(_pair_time(main, main, us) for _ in range(2))
    pair = min(_pair_time(main, stream, us) for _ in range(3))
    return pair < 0.75 * serial


def init_streams(force=False):
    """Creates the side streams. The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES (4) hardware
    queues, and torch's stream pool + RCCL create dozens: if the filter-gradient stream lands on the compute stream's queue
    the two backward chains serialise silently (measured under torch.distributed: 831 instead of 884 img/s). So the candidates
    are PROBED (two idle kernels, _runs_beside) and a stream that really runs beside the current one is taken. Call it after
    torch.distributed is initialised (DataParallel does) and when a model is prepared for training."""
    global _COPY_STREAM, _SIDE_FILTER, _WGRAD_STREAM, _SORT_STREAM
    main = torch.cuda.current_stream()
    if _WGRAD_STREAM is None or force:
        cands, good = [], []
        while len(good) < 4 and len(cands) < 32:          # torch hands out pool streams round-robin: a few tries reach every queue
            st = torch.cuda.Stream()
            cands.append(st)
            if _runs_beside(st, main):
                good.append(st)
        _WGRAD_STREAM = good[0] if good else cands[0]
        # filter transforms / tap sort: beside the compute stream too, and preferably not on the filter-gradient queue
        rest = [s for s in good[1:] if _runs_beside(s, _WGRAD_STREAM)] or good[1:] or cands[1:]
        _SIDE_FILTER = rest[0]
        _SORT_STREAM = rest[1] if len(rest) > 1 else rest[0]
        # uploads of targets / the early copy of the costs: off the compute stream's queue as well (behind it they would wait
        # for whatever kernels are queued there)
        _COPY_STREAM = rest[2 % len(rest)]
        SIDE_STREAMS[:] = rest


def upload_async(pinned):
    """H2D copy of a pinned host tensor on a dedicated copy stream: a target array issued on the compute stream
    would sit between two kernels and stall them for the PCIe time (4-8 MB = 0.1-0.3 ms per step). Returns
    (device tensor, event); the consumer calls wait_upload(event) right before the first kernel that reads it."""
    global _COPY_STREAM
    if _COPY_STREAM is None:
        init_streams()
    with torch.cuda.stream(_COPY_STREAM):
        dev = pinned.cuda(non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(_COPY_STREAM)
    dev.record_stream(torch.cuda.current_stream())
    return dev, ev


def wait_upload(ev):
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


def wait_stream(idle=None, period=0.0):
    """host wait for everything queued on the current stream. Polls an event instead of a blocking
    hipStreamSynchronize: the interrupt-driven wait was measured to wake up 10-30 ms late now and then on the
    GPU box, which is a third of a training step; the mid-step RoI hand-off is latency critical.
    idle / period: a callable the host runs every `period` seconds while it polls (the RoI hand-off keeps its native call warm)"""
    ev = torch.cuda.Event()
    ev.record()
    if idle is None or period <= 0:
        while not ev.query():
            pass
        return
    import time
    last = time.perf_counter()
    while not ev.query():
        now = time.perf_counter()
        if now - last >= period:
            idle()
            last = time.perf_counter()


def _last_igemm_name():
    """name of the instantiation the C side picked for the launch just issued (matches rocprofv3's kernel names:
    igemm_kernel<MODE, BM, BN, 2, 2, NBUF>)"""
    import ctypes
    v = [ctypes.c_int() for _ in range(5)]
    _L().denet_conv_last_config(*[ctypes.byref(x) for x in v])
    mode, bm, bn, nbuf, _ = [x.value for x in v]
    return "igemm_kernel<%d, %d, %d, 2, 2, %d>" % (mode, bm, bn, nbuf)


def _conv_flops(g, logical=None):
    """algorithmic FLOPs = 2 * MACs over the LOGICAL channel counts (padding channels / taps do not count)"""
    N, H, W, C, K, R, S, s_real, stride, pad, OH, OW = g
    if logical is not None:
        C, K = logical
    return 2.0 * N * OH * OW * K * R * s_real * C


def conv_geom(x_shape, w_shape, stride, pad, s_real=None):
    N, H, W, C = x_shape
    K, R, S, Cw = w_shape
    assert Cw == C, (x_shape, w_shape)
    s_real = S if s_real is None else s_real
    OH = (H + 2 * pad - R) // stride + 1
    OW = (W + 2 * pad - s_real) // stride + 1
    return N, H, W, C, K, R, S, s_real, stride, pad, OH, OW


# DENET_AUTOTUNE=0 ignores the committed decisions and the Winograd / fused algorithms: built-in launch heuristics, direct kernels
AUTOTUNE = os.environ.get("DENET_AUTOTUNE", "1") != "0"
# Measuring is an EXPLICIT act (DENET_TUNE=1, or ops.MEASURE = True as tools/tune.py does): only then a (pass, geometry) the
# committed file does not cover has its launch configuration (denet_conv_tune) and its algorithm timed on the first call. The
# product default never measures: such a geometry runs static_policy's algorithm on the C side's launch heuristics, so two
# processes always run the same kernels and training is reproducible run to run at ANY size (a first-step timing race ended
# differently per box: round-5 verdict, weak 1). tools/tune.py extends the file for a new geometry.
MEASURE = os.environ.get("DENET_TUNE", "0") == "1"
_TUNED = set()


# ---- persisted launch configurations ------------------------------------------------------------------------------------
# denet_amd/tuned/gfx950.json (written by tools/tune.py on an MI355X, committed) holds, per geometry, the measured kernel
# configuration of every pass and the direct / Winograd decision. It is loaded with the library: the geometries it covers are
# never measured again, so two processes run the SAME kernels (bench.py, the rocprofv3 profiles and the PMC passes
# describe one launch population). DENET_TUNE_CACHE=<path> selects another file, DENET_TUNE_CACHE=0 ignores it.
TUNE_CACHE = os.environ.get("DENET_TUNE_CACHE", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950.json"))
_TUNE_LOADED = False


def _geom_of_record(r):
    mode, N, H, W, C, K, R, S, s_real, stride, pad = r[:11]
    OH = (H + 2 * pad - R) // stride + 1
    OW = (W + 2 * pad - s_real) // stride + 1
    return mode, (N, H, W, C, K, R, S, s_real, stride, pad, OH, OW)


def load_tuned(path=None):
    """imports a configuration file; returns the number of kernel records"""
    import ctypes
    import json
    path = path or TUNE_CACHE
    with open(path) as f:
        d = json.load(f)
    rec = d.get("kernels", [])
    if rec:
        flat = (ctypes.c_int * (14 * len(rec)))(*[v for r in rec for v in r])
        check(_L().denet_tune_import(flat, len(rec)), "tune_import")
    for r in rec:
        if r[0] <= 2:
            _TUNED.add(_geom_of_record(r))
    for mode, g, tile in d.get("winograd", []):
        if _tile_allowed(int(mode), int(tile)):  # DENET_WINOGRAD / DENET_WINO2F restrict the algorithms: excluded entries
            _WINO[(int(mode), tuple(int(v) for v in g))] = int(tile)         # are decided afresh
    return len(rec)


def save_tuned(path, meta=None):
    import ctypes
    import json
    n = _L().denet_tune_export(None, 0)
    buf = (ctypes.c_int * (14 * max(n, 1)))()
    _L().denet_tune_export(buf, n)
    rec = sorted([list(buf[i * 14:(i + 1) * 14]) for i in range(n)])
    wino = sorted([[m, list(g), t] for (m, g), t in _WINO.items()])
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"meta": meta or {}, "kernels": rec, "winograd": wino}, f, separators=(",", ":"))
    return n


def _load_tuned_once():
    global _TUNE_LOADED
    if _TUNE_LOADED:
        return
    _TUNE_LOADED = True
    if AUTOTUNE and TUNE_CACHE not in ("0", "") and os.path.exists(TUNE_CACHE):
        load_tuned(TUNE_CACHE)


def _tune_first(mode, g, a, b, bias, add, out, ws):
    """True if this call was served by the tuner (which leaves the pass's result in `out`)"""
    _load_tuned_once()
    if not AUTOTUNE or not MEASURE or POLICY is not None or PROFILE is not None or (mode, g) in _TUNED:
        return False
    _TUNED.add((mode, g))
    check(_L().denet_conv_tune(mode, ptr(a), ptr(b), ptr(bias), ptr(add), ptr(out), ptr(ws), ws.numel() if ws is not None
                               else 0, *g, stream_ptr()), "conv_tune")
    return True


# OPT-IN (DENET_HEAD_BF16X3=1, or ops.HEAD_BF16X3 = True): the 1x1 stride-1 convolutions with >= 512 input channels (the detection
# head) as 3-term bf16-split GEMMs on the bf16 matrix cores (csrc/gemm3b.hip): ~1e-6 relative error instead of the exact fp32
# FMA chain, 2x the speed. Never on by default; bench.py reports it under its own key.
HEAD_BF16X3 = os.environ.get("DENET_HEAD_BF16X3", "0") == "1"


# The second stage of a batch norm's reductions inside the launch that writes the partial sums (csrc/bn_final.h, denet_bn_final_arm_*):
# the last workgroup of the producing convolution pass reduces the rows, and the 84 bn_stats_final / bn_bwd_final launches of a
# DeNet-34 step - 5 us of work each, but 25-37 us on the backward sweep's critical chain, where such a launch waits for a CU slot
# beside the other stream's matrix kernels - leave the streams. Bit-identical to the separate launches.
# OFF by default: measured (MI355X, A / B on one box, tools/exp/ab_fold.sh) 1 021-1 027 img/s with both kinds folded, 1 013-1 046 with
# the backward sums only, 1 047-1 055 without - the last workgroup's serial tail (ticket round trip under 256 simultaneous arrivals
# + the dependent row loads behind the launch's own store burst) costs a producing kernel +5...23 us, the launch it replaces 5 us,
# and what that launch waits for a CU slot inside the step is time the other chain uses anyway. DENET_BN_FINAL_FOLD=3 (bit 0 forward
# statistics, bit 1 backward sums) / set_final_fold() switch it on; the bit-identity tests run it.
FINAL_FOLD = os.environ.get("DENET_BN_FINAL_FOLD", "0") not in ("0", "")


def set_final_fold(bits):
    """which batch-norm reductions the producing passes finish themselves (bit 0 forward statistics, bit 1 backward sums; 0 = the
    separate final launches, the default); returns the previous setting"""
    global FINAL_FOLD
    old = int(_L().denet_bn_final_mode(int(bits)))
    FINAL_FOLD = bool(bits)
    return old
FINAL_COUNT = [0, 0]        # forward statistics / backward sums finished by their producers (tests, bench)


class BnFinal:
    """the batch norm whose reduction the next producing pass may finish itself. kind 1 (forward statistics): run_mean / run_stdinv
    are updated, save_mean / save_invstd come back; kind 2 (backward sums): dgamma / dbeta are written, coef [2][C] comes back.
    holder: a dict that keeps the counters alive (one producing pass at a time uses them). arm() right before the producing call,
    disarm() right after: self.taken says whether the pass took the final over."""

    def __init__(self, kind, M, C, holder, momentum=0.0, eps=0.0, run_mean=None, run_stdinv=None, dgamma=None, dbeta=None):
        self.kind, self.M, self.C = int(kind), int(M), int(C)
        self.momentum, self.eps, self.run_mean, self.run_stdinv = float(momentum), float(eps), run_mean, run_stdinv
        self.dgamma, self.dbeta = dgamma, dbeta
        self.holder = holder
        self.taken = False
        self.save_mean = self.save_invstd = self.coef = None

    def arm(self):
        cnt = self.holder.get("bnf_counters")
        if cnt is None:
            cnt = self.holder["bnf_counters"] = torch.zeros(64, dtype=torch.int32, device="cuda")
        if self.kind == 1:
            self.save_mean, self.save_invstd = empty(self.C), empty(self.C)
            check(_L().denet_bn_final_arm_stats(self.M, self.C, self.momentum, self.eps, ptr(self.run_mean), ptr(self.run_stdinv),
                                                ptr(self.save_mean), ptr(self.save_invstd), ptr(cnt), 64), "bn_final_arm_stats")
        else:
            self.coef = empty(2 * self.C)
            check(_L().denet_bn_final_arm_sums(self.M, self.C, ptr(self.dgamma), ptr(self.dbeta), ptr(self.coef), ptr(cnt), 64),
                  "bn_final_arm_sums")

    def disarm(self):
        self.taken = bool(_L().denet_bn_final_disarm())
        if self.taken:
            FINAL_COUNT[self.kind - 1] += 1
        return self.taken


class _armed:
    """with _armed(fin): <one producing C call>   (fin None or the fold switched off: nothing happens)"""

    def __init__(self, fin):
        self.fin = fin if (fin is not None and FINAL_FOLD) else None

    def __enter__(self):
        if self.fin is not None:
            self.fin.arm()
        return self.fin

    def __exit__(self, *a):
        if self.fin is not None:
            self.fin.disarm()
        return False


def _stats_result(st, rows, fin):
    """what a producing forward pass leaves in cache["bn_stats"]: (partial sums, rows[, the BnFinal it finished])"""
    if rows <= 0:
        return None
    return (st, rows, fin) if (fin is not None and fin.taken) else (st, rows)


def _conv_wino_fwd_linked(link, g, tile, w, bias, add, out, cache, bn_stats):
    """conv_fwd's Winograd branch with the input transform that evaluates the batch norm in front (denet_conv_wino_fwd_fold)"""
    import ctypes
    N, H, W, C, K, R, S, s_real, stride, pad, OH, OW = g
    y = out if out is not None else empty(N, OH, OW, K)
    act = torch.empty_like(link.x)
    st = None
    rows = ctypes.c_int(0)
    if bn_stats:
        nrows = max((N * OH * OW + 127) // 128, (N * (OH // 2 + 1) * (OW // 2 + 1) * (K // 4) + 255) // 256,
                    N * ((OH + 15) // 16) * ((OW + 15) // 16))
        st = cache.get("bn_stats_buf")
        if st is None or st.numel() < nrows * 2 * K:
            st = cache["bn_stats_buf"] = torch.empty(nrows * 2 * K, dtype=torch.float64, device="cuda")
        cache["bn_stats"] = None
    cache["fwd_tile"] = tile
    u = _cached_u(cache, 0, tile)
    v_keep = None
    if _decided(2, g) == tile:               # the filter gradient of this layer reuses the transformed input
        v_keep = cache.get("V")
        nv = (tile + 2) * (tile + 2) * (N * -(-H // tile) * -(-W // tile)) * C
        if v_keep is None or v_keep.numel() != nv:
            v_keep = cache["V"] = torch.empty(nv, dtype=torch.float32, device="cuda")
        cache["V_tile"] = tile
    ws = _wino_ws(tile, N, H, W, C, K)
    bn = link.c_struct(act)
    with _armed(cache.get("bn_final") if st is not None else None) as fin:
        check(_L().denet_conv_wino_fwd_fold(ctypes.byref(bn), ptr(w), ptr(u), ptr(v_keep), ptr(bias), ptr(add), ptr(y), 0, ptr(st),
                                            st.numel() * 8 if st is not None else 0, ctypes.byref(rows), ptr(ws), ws.numel(), tile,
                                            N, H, W, C, K, stream_ptr()), "conv_wino_fwd_fold")
    if st is not None:
        cache["bn_stats"] = _stats_result(st, rows.value, fin)
    link.result = act
    LINK_COUNT[0] += 1
    return y


def conv_backward_linked(link, x, w, w_shape, add, dw_out, cache, stride=1, pad=1, s_real=None, logical=None, sums=None):
    """Data gradient AND filter gradient of a 3x3 convolution whose output gradient is the (unwritten) result of a batch norm's
    backward pointwise pass (BnLink): one transform kernel on the compute stream evaluates it and writes both transformed
    tensors (denet_conv_wino_dgrad_fold), the filter-gradient products follow on the second stream (denet_conv_wino_wgrad_dm).
    Returns dx, or None when this layer's passes are not both Winograd passes of one tile (the caller materialises the gradient
    and takes the ordinary path)."""
    import ctypes
    if not (int(BWD_SUMS) & 1):
        sums = None                  # DENET_BN_BWD_SUMS bit 0: the Winograd passes leave the backward reductions to the batch norm
    g = conv_geom(x.shape, w_shape, stride, pad, s_real)
    N, H, W, C, K = g[0], g[1], g[2], g[3], g[4]
    tile = _decided(1, g)
    if not (LINK_BN and tile in (2, 4) and _decided(2, g) == tile and not _bf16x3_geom(g)):
        return None
    if PROFILE is not None:          # two implicit-GEMM launches follow, in this order: data-gradient, filter-gradient products
        PROFILE.add(_conv_flops(g, logical) / _WINO_GAIN[tile])
        PROFILE.add(_conv_flops(g, logical) / _WINO_GAIN[tile])
    if cache is not None:
        cache["dgrad_tile"] = tile
    T = N * -(-H // tile) * -(-W // tile)
    dm = torch.empty((tile + 2) * (tile + 2) * T * K, dtype=torch.float32, device="cuda")
    dx = empty(N, H, W, C)
    u = _cached_u(cache, 1, tile) if cache is not None else None
    ws = _wino_ws(tile, N, H, W, C, K)
    ev = cache.get("prep_event") if cache is not None else None
    if ev is None:
        ev = torch.cuda.Event()
        ev.record()                                   # creates the handle the native call records on
        if cache is not None:
            cache["prep_event"] = ev
    bn = link.c_struct(link.out)
    rows = ctypes.c_int(0)
    sb = sums.buffer(cache, (T * (C // 4) + 255) // 256, C) if sums is not None else None
    so = sums.c_struct() if sums is not None else None
    with _armed(sums.final if sums is not None else None) as fin:
        check(_L().denet_conv_wino_dgrad_fold(ctypes.byref(bn), ptr(dm), ptr(w), ptr(u), ptr(add), ptr(dx),
                                              ctypes.byref(so) if so is not None else None, ptr(sb), sb.numel() * 8 if sb is not None else 0,
                                              ctypes.byref(rows), ptr(ws), ws.numel(), tile, N, H, W, C, K, ev.cuda_event, stream_ptr()),
              "conv_wino_dgrad_fold")
    if sums is not None:
        sums.done(sb, rows.value, fin)
    v = None
    if cache is not None and cache.get("V_tile") == tile:
        v = cache.get("V")
        cache["V_tile"] = None
    global _ON_WGRAD_STREAM
    if WGRAD_STREAM and _WGRAD_STREAM is None:
        init_streams()
    side = _WGRAD_STREAM if (WGRAD_STREAM and (PROFILE is None or not PROFILE.alone)) else None      # a live kernel profile runs every kernel alone
    if side is not None:
        side.wait_event(ev)                           # only the transform kernel: the products of the two chains run side by side
        dm.record_stream(side)
        with torch.cuda.stream(side):
            _ON_WGRAD_STREAM = True
            try:
                _wgrad_dm(x, dm, v, dw_out, tile, N, H, W, C, K)
            finally:
                _ON_WGRAD_STREAM = False
    else:
        _wgrad_dm(x, dm, v, dw_out, tile, N, H, W, C, K)
    LINK_COUNT[1] += 1
    return dx


def _wgrad_dm(x, dm, v, dw, tile, N, H, W, C, K):
    ws = _wino_ws(tile, N, H, W, C, K)
    sws = WS.get("wgrad", WGRAD_WS_BYTES)
    check(_L().denet_conv_wino_wgrad_dm(ptr(x), ptr(dm), ptr(v), ptr(dw), ptr(ws), ws.numel(), ptr(sws), sws.numel(), tile, N, H, W,
                                        C, K, stream_ptr()), "conv_wino_wgrad_dm")


def _bf16x3_geom(g):
    N, H, W, C, K, R, S, s_real, stride, pad, OH, OW = g
    return HEAD_BF16X3 and R == 1 and S == 1 and stride == 1 and pad == 0 and C >= 512 and C % 128 == 0 and K % 128 == 0 \
        and (N * H * W) % 32 == 0


def _gemm_bf16x3(a, b, bias, out, M, N, K):
    check(_L().denet_gemm_bf16x3_nt(ptr(a), ptr(b), ptr(bias), ptr(out), M, N, K, stream_ptr()), "gemm_bf16x3_nt")
    return out


def _transpose(src, R, C, out=None):
    dst = out if out is not None else empty(C, R)
    check(_L().denet_transpose_f32(ptr(src), ptr(dst), R, C, stream_ptr()), "transpose_f32")
    return dst


def conv_fwd(x, w, bias=None, add=None, stride=1, pad=0, s_real=None, out=None, logical=None, cache=None, relu=False,
             bn_stats=False, link=None, up=None):
    """bn_stats (training, the layer behind is a batch norm): the epilogue of the pass also writes the per-channel sums of y;
    cache["bn_stats"] = (partial sums tensor, rows) for bn_fwd_train(pre=...), or None when the chosen kernel cannot.
    link (BnLink, x = None): the input is the output of a batch norm whose pointwise pass has not run; a Winograd pass evaluates
    it inside its input transform, every other implementation materialises it first. link.result is the activation afterwards."""
    if link is not None:
        g = conv_geom(link.x.shape, w.shape, stride, pad, s_real)
        tile = _decided(0, g)
        if not (tile in (2, 4) and cache is not None and cache.get("train") and not relu and not _bf16x3_geom(g)):
            x = link.materialise()             # direct / fused-64 / undecided implementations read the tensor itself
            link = None
    if link is not None:
        if PROFILE is not None:
            PROFILE.add(_conv_flops(g, logical) / _WINO_GAIN[tile])
        return _conv_wino_fwd_linked(link, g, tile, w, bias, add, out, cache, bn_stats)
    if up is not None:
        # up (UpLink, x = None): the input is a pool-inverse layer's output that has not been written. A training pass on the
        # un-fused Winograd kernels (already decided for this geometry) reads the small tensor in its input transform; every other
        # case writes the up-sampled tensor first
        gu = conv_geom(up.shape, w.shape, stride, pad, s_real)
        if not (_decided(0, gu) in (2, 4) and ((0, gu) in _TUNED or not AUTOTUNE or not MEASURE or POLICY is not None) and cache is not None and cache.get("train") and bn_stats
                and not relu and not _bf16x3_geom(gu)):
            x = up.materialise()
            up = None
    g = conv_geom(up.shape if up is not None else x.shape, w.shape, stride, pad, s_real)
    N, H, W, C, K, R, S, s_real, stride, pad, OH, OW = g
    y = out if out is not None else empty(N, OH, OW, K)
    if add is None and not relu and _bf16x3_geom(g):
        if cache is not None:
            cache["fwd_tile"] = 0
            cache["bn_stats"] = None               # the batch norm behind measures its own statistics
        return _gemm_bf16x3(x, w, bias, y, N * H * W, K, C)
    _tune_first(0, g, x, w, bias, add, y, None)
    st = None
    if bn_stats and cache is not None and not relu:
        # rows of partial sums: direct tiles of 128 pixels, the F(2x2) output transform, the fused 64-channel kernel (one
        # row per 16x16-pixel block) - whichever implementation runs, the buffer holds its rows
        rows = max((N * OH * OW + 127) // 128, (N * (OH // 2 + 1) * (OW // 2 + 1) * (K // 4) + 255) // 256,
                   N * ((OH + 15) // 16) * ((OW + 15) // 16), N * ((OH + 7) // 8) * ((OW + 31) // 32))
        st = cache.get("bn_stats_buf")
        if st is None or st.numel() < rows * 2 * K:
            st = cache["bn_stats_buf"] = torch.empty(rows * 2 * K, dtype=torch.float64, device="cuda")
        cache["bn_stats"] = None

    def direct(final=False):
        if final and st is not None:
            import ctypes
            rows = ctypes.c_int(0)
            with _armed(cache.get("bn_final")) as fin:
                check(_L().denet_conv_fwd_stats(ptr(x), ptr(w), ptr(bias), ptr(add), ptr(y), ptr(st), st.numel() * 8,
                                                ctypes.byref(rows), *g, stream_ptr()), "conv_fwd_stats")
            cache["bn_stats"] = _stats_result(st, rows.value, fin)
            return
        check(_L().denet_conv_fwd_act(ptr(x), ptr(w), ptr(bias), ptr(add), ptr(y), int(relu), *g, stream_ptr()), "conv_fwd")

    # the implementation is fixed on the first call; the timed candidates scribble over `y`, so the call always ends
    # with a launch of the chosen one (run-to-run determinism)
    tile = _wino_tile(0, g, direct, lambda t: conv_wino_fwd(x, w, bias, add, out=y, tile=t, relu=relu))
    # INFERENCE (no filter gradient will want the transformed input of this pass, the batch norms are folded into plain tensors):
    # the tuned file may name another algorithm for the forward pass alone - mode 3 entries (the tile-parallel fused F(4x4) kernel
    # on the many-channel layers, where training keeps the component-walk kernel because its V feeds the filter gradient)
    if cache is not None and not cache.get("train") and (3, g) in _WINO and _tile_allowed(0, _WINO[(3, g)]):
        t3 = _WINO[(3, g)]
        if (t3 == FUSED4 and conv_wino4t_ok(0, g)) or (t3 in (2, 4) and conv_wino_ok(g, t3)) or t3 == 0:
            tile = t3
    if tile:
        if PROFILE is not None:
            PROFILE.add(_conv_flops(g, logical) / _WINO_GAIN[tile])     # the FLOPs its batched GEMM really executes
        u = v_keep = None
        if cache is not None:
            cache["fwd_tile"] = tile
            u = _cached_u(cache, 0, tile)
            if u is None and not cache.get("train"):
                # inference: the filters do not change between calls - transform them once per weights version
                ent = cache.get("u_test")
                if ent is None or ent[0] != tile or ent[2] != WEIGHTS_VERSION or ent[3] != w.data_ptr():
                    uf = conv_wino4t_filter(w, dgrad=False) if tile == FUSED4 else conv_wino_filter(w, _filter_tile(tile), dgrad=False)
                    ent = cache["u_test"] = (tile, uf, WEIGHTS_VERSION, w.data_ptr())
                u = ent[1]
            # the filter gradient of this layer uses the same transformed input when it runs with the same tile
            if cache.get("train") and _decided(2, g) == tile:
                v_keep = cache.get("V")
                nv = (tile + 2) * (tile + 2) * (N * -(-H // tile) * -(-W // tile)) * C
                if v_keep is None or v_keep.numel() != nv:
                    v_keep = cache["V"] = torch.empty(nv, dtype=torch.float32, device="cuda")
                cache["V_tile"] = tile
        if up is not None and st is None:
            x, up = up.materialise(), None
        return conv_wino_fwd(x, w, bias, add, out=y, tile=tile, u=u, v_keep=v_keep, relu=relu, stats=(st, cache) if st is not None else None,
                             up=up)
    assert up is None
    if cache is not None:
        cache["fwd_tile"] = 0
    if PROFILE is not None:
        PROFILE.add(_conv_flops(g, logical))
    direct(final=True)
    return y


# Winograd F(2x2,3x3) / F(4x4,3x3) are alternative implementations of the eligible 3x3 layers; which one runs is measured
# once per (pass, geometry) right after the direct kernel has been tuned. DENET_WINOGRAD=0 disables the paths,
# DENET_WINOGRAD=2 allows only F(2x2,3x3).
WINOGRAD = int(os.environ.get("DENET_WINOGRAD", "4"))
WINO_RAGGED = os.environ.get("DENET_WINO_RAGGED", "1") != "0"
_WINO = {}
# A third alternative for the 64-input-channel layers: FUSED2, F(2x2,3x3) with the transforms and the products in one kernel
# (csrc/wino2f.hip). DENET_WINO2F: bit 0 allows it for the forward pass, bit 1 for the data gradient, bit 2 for the filter
# gradient (64 -> 64 channels only).
FUSED2 = 22
WINO2F = int(os.environ.get("DENET_WINO2F", "7"))
# A fourth: FUSED4, F(4x4,3x3) tile-parallel with BOTH transforms and the 36 products in one kernel (csrc/wino4t.hip): x -> y
# only, like FUSED2, at 1.78 times fewer products. Forward pass and data gradient (DENET_WINO4T bits 0 / 1); reduction channels
# a multiple of 16, written channels of 64, H and W multiples of 4. Chosen by the tuned file (or a policy), never by static_policy.
FUSED4 = 44
WINO4T = int(os.environ.get("DENET_WINO4T", "3"))
_WINO_GAIN = {2: 2.25, 4: 4.0, FUSED2: 2.25, FUSED4: 4.0}      # direct multiplications / Winograd multiplications


# POLICY: a callable (mode, geometry) -> tile that DECIDES the implementation of a pass that has no entry in _WINO yet (at the
# benchmark geometries the committed tuned file decides). None = the default: static_policy below, or - only with MEASURE on -
# timing the candidates on the first call. With a policy set no launch configuration is measured either (the C side's
# heuristics apply).
POLICY = None


def static_policy(mode, g):
    """the fused 64-channel kernels where they apply, else F(4x4), else F(2x2), else the direct kernel"""
    if _tile_allowed(mode, FUSED2) and conv_wino2f_ok(mode, g):
        return FUSED2
    for t in (4, 2):
        if t <= WINOGRAD and conv_wino_ok(g, t):
            return t
    return 0


def _decided(mode, g):
    """the implementation fixed for this pass (0 direct, 2 / 4 Winograd tile, FUSED2), None while undecided"""
    _load_tuned_once()
    use = _WINO.get((mode, g))
    if use is None:
        policy = POLICY if POLICY is not None else (static_policy if AUTOTUNE and not MEASURE else None)
        if policy is not None:
            use = _WINO[(mode, g)] = int(policy(mode, g))
    return use


def _tile_allowed(mode, tile):
    if tile == FUSED2:
        return WINOGRAD >= 2 and mode in (0, 1, 2) and bool((WINO2F >> mode) & 1)
    if tile == FUSED4:
        return WINOGRAD >= 4 and mode in (0, 1, 3) and bool((WINO4T >> (mode & 1)) & 1)       # (mode 3: the inference forward pass)
    return tile <= WINOGRAD


def _filter_tile(tile):
    """the Winograd tile whose transformed filters the algorithm consumes"""
    return 2 if tile == FUSED2 else (4 if tile == FUSED4 else tile)


def conv_wino4t_ok(mode, g):
    """geometry the tile-parallel fused F(4x4,3x3) kernel covers for the forward pass (mode 0) / the data gradient (1)"""
    N, H, W, C, K, R, S, s_real, stride, pad, OH, OW = g
    if not (mode in (0, 1) and R == 3 and S == 3 and s_real == 3 and stride == 1 and pad == 1):
        return False
    red, out = (C, K) if mode == 0 else (K, C)
    return bool(_L().denet_conv_wino4t_ok(N, H, W, red, out))


def conv_wino2f_ok(mode, g):
    """geometry the fused F(2x2,3x3) kernels cover for the forward pass (mode 0) / the data gradient (1) / the filter
    gradient (2)"""
    N, H, W, C, K, R, S, s_real, stride, pad, OH, OW = g
    if not (R == 3 and S == 3 and s_real == 3 and stride == 1 and pad == 1):
        return False
    if mode == 2:
        return bool(_L().denet_conv_wino2f_wgrad_ok(N, H, W, C, K))
    ci, co = (C, K) if mode == 0 else (K, C)
    return bool(_L().denet_conv_wino2f_ok(N, H, W, ci, co))


def _time_ms(fn, reps=3):
    fn()
    best = 1e30
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        best = min(best, s.elapsed_time(e))
    return best


def conv_wino_ok(g, tile=2):
    """geometry the Winograd F(tile x tile, 3x3) path covers: 3x3, stride 1, pad 1, channels multiples of 32. A map that is no
    multiple of the tile is covered by ceil(H / tile) x ceil(W / tile) tiles (zeros read, results dropped beyond the map: the 14x14 and
    7x7 maps of ResNet-34 at 224x224 run F(4x4) on 16x16 / 8x8 tile grids); DENET_WINO_RAGGED=0 restores the multiples-only rule"""
    N, H, W, C, K, R, S, s_real, stride, pad, OH, OW = g
    return R == 3 and S == 3 and s_real == 3 and stride == 1 and pad == 1 and (WINO_RAGGED or (H % tile == 0 and W % tile == 0)) \
        and H >= tile and W >= tile and C % 32 == 0 and K % 32 == 0


def _wino_tile(mode, g, direct, wino):
    """decides (once) between the direct kernel (0) and the Winograd tiles (2, 4) for this pass by timing them;
    wino(tile) runs the pass with the given tile"""
    key = (mode, g)
    use = _decided(mode, g)
    if use is None:
        tiles = [t for t in (2, 4) if t <= WINOGRAD and conv_wino_ok(g, t)] if AUTOTUNE else []
        if AUTOTUNE and _tile_allowed(mode, FUSED2) and conv_wino2f_ok(mode, g):
            tiles.append(FUSED2)
        if AUTOTUNE and _tile_allowed(mode, FUSED4) and conv_wino4t_ok(mode, g) and (g[3] if mode == 0 else g[4]) == 64:
            tiles.append(FUSED4)         # (measured only where the fused F(2x2) kernel is the alternative: 64 reduction channels)
        if not tiles:
            use = 0
        elif PROFILE is not None:
            return 0                     # undecided while a profile is being recorded: direct kernel, decide later
        else:
            N, H, W, C, K = g[0], g[1], g[2], g[3], g[4]
            sws = WS.get("wgrad", WGRAD_WS_BYTES)
            best, use = 0.97 * _time_ms(direct), 0
            for t in tiles:
                if t not in (FUSED2, FUSED4):    # the batched products of the un-fused passes have configurations of their own
                    ws = WS.get("wino", _L().denet_conv_wino_workspace_bytes(t, N, H, W, C, K))
                    check(_L().denet_conv_wino_tune(ptr(ws), ws.numel(), ptr(sws), sws.numel(), t, N, H, W, C, K, stream_ptr()),
                          "conv_wino_tune")
                ms = _time_ms(lambda: wino(t))
                if ms < best:
                    best, use = ms, t
        _WINO[key] = use
    return use


def _cached_u(cache, dgrad, tile):
    """transformed filters prepared ahead by wino_prefetch_filters (None: the call transforms them itself)"""
    ent = cache.get(("u", dgrad))
    if ent is None or ent[0] != tile or not ent[2]:
        return None
    wait_upload(cache.get(("u_event", dgrad)))
    ent[2] = False                            # valid for one step: the solver changes the weights
    return ent[1]


def _cached_ws2(cache):
    """packed filter of a 3x3 stride-2 layer's data gradient prepared ahead by wino_prefetch_filters (None: the caller packs)"""
    ent = cache.get("ws2") if cache is not None else None
    if ent is None or not ent[1]:
        return None
    wait_upload(cache.get("ws2_event"))
    ent[1] = False                            # valid for one step: the solver changes the weights
    return ent[0]


def conv_dgrad_s2_pack(w, out=None):
    """w [K][3][3][C] -> [K/16][9][C][16] for denet_conv_dgrad_s2"""
    K, _, _, C = w.shape
    pk = out if out is not None else empty(9 * K * C)
    check(_L().denet_conv_dgrad_s2_pack(ptr(w), ptr(pk), C, K, stream_ptr()), "conv_dgrad_s2_pack")
    return pk


def _cached_wt(cache):
    """transposed filter of a large 1x1 layer prepared ahead by wino_prefetch_filters (None: the caller transposes)"""
    ent = cache.get("wt") if cache is not None else None
    if ent is None or not ent[1]:
        return None
    wait_upload(cache.get("wt_event"))
    ent[1] = False                            # valid for one step: the solver changes the weights
    return ent[0]


def wino_prefetch_filters(caches_and_weights, after=None):
    """For every convolution layer that runs Winograd passes: transform its filters for the forward and the data-gradient
    pass on a side stream, right at the start of a training step (the ~60 small launches leave the critical path)."""
    global _SIDE_FILTER
    todo = [(c, w) for c, w in caches_and_weights if c.get("fwd_tile") or c.get("dgrad_tile")]
    tr = [(c, w) for c, w in caches_and_weights if c.get("dgrad_1x1t") or c.get("dgrad_t")]
    s2 = [(c, w) for c, w in caches_and_weights if c.get("dgrad_s2")]
    if not todo and not tr and not s2:
        return
    if _SIDE_FILTER is None:
        init_streams()
    # ordered behind the solver update of the weights only (after: the event recorded when the step began), not behind the
    # kernels the forward pass has queued since - the ~300 small launches run beside the stem convolution; the forward
    # filters of the first layers are transformed first and every GROUP of layers signals its own event, so that the first
    # Winograd layer does not wait for the last layer's filters (round 3: with one event behind all launches the first
    # 64-channel block started ~1 ms late)
    if after is not None:
        _SIDE_FILTER.wait_event(after)
    else:
        _SIDE_FILTER.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(_SIDE_FILTER):
        def run(items, dgrad, key):
            for c, w in items:
                tile = c.get(key)
                if not tile:
                    continue
                ent = c.get(("u", dgrad))
                ft = _filter_tile(tile)
                if ent is None or ent[0] != tile:
                    K, _, _, C = w.shape
                    ent = c[("u", dgrad)] = [tile, torch.empty((ft + 2) * (ft + 2) * K * C, dtype=torch.float32, device="cuda"), False]
                if tile == FUSED4:
                    conv_wino4t_filter(w, dgrad, out=ent[1])
                else:
                    conv_wino_filter(w, ft, dgrad, out=ent[1])
                ent[2] = True
            ev = torch.cuda.Event()
            ev.record(_SIDE_FILTER)
            for c, _ in items:
                c[("u_event", dgrad)] = ev
        # forward filters in layer order, in groups of 8 layers; then the data-gradient filters (needed much later)
        for i in range(0, len(todo), 8):
            run(todo[i:i + 8], 0, "fwd_tile")
        run(todo, 1, "dgrad_tile")
        if tr:
            # the transposed filters of the large 1x1 layers' data-gradient products (conv_dgrad)
            for c, w in tr:
                K = w.shape[0]
                C = w.numel() // K                 # (R * S * C columns: wt [R][S][C][K] for denet_conv_dgrad_t; [C][K] for a 1x1 layer)
                ent = c.get("wt")
                if ent is None or ent[0].numel() != K * C:
                    ent = c["wt"] = [torch.empty(C, K, dtype=torch.float32, device="cuda"), False]
                _transpose(w, K, C, out=ent[0])
                ent[1] = True
            ev = torch.cuda.Event()
            ev.record(_SIDE_FILTER)
            for c, _ in tr:
                c["wt_event"] = ev
        if s2:
            # the packed filters of the 3x3 stride-2 layers' data gradients (conv_dgrad -> csrc/dgrad_s2.hip)
            for c, w in s2:
                ent = c.get("ws2")
                if ent is None or ent[0].numel() != w.numel():
                    ent = c["ws2"] = [torch.empty(w.numel(), dtype=torch.float32, device="cuda"), False]
                conv_dgrad_s2_pack(w, out=ent[0])
                ent[1] = True
            ev = torch.cuda.Event()
            ev.record(_SIDE_FILTER)
            for c, _ in s2:
                c["ws2_event"] = ev


_SIDE_FILTER = None


_WGRAD_STREAM = None
_ON_WGRAD_STREAM = False
WGRAD_STREAM = os.environ.get("DENET_WGRAD_STREAM", "1") != "0"
# a layer whose data-gradient GEMM is at least this large runs it BEFORE its filter gradient is queued (layer/convolution.py)
DGRAD_FIRST_GFLOP = float(os.environ.get("DENET_DGRAD_FIRST_GFLOP", "100"))
# a 1x1 stride-1 layer whose data-gradient GEMM is at least this large runs it as a forward product over the transposed filter
# (conv_dgrad; 0 = never)
DGRAD_1X1T_GFLOP = float(os.environ.get("DENET_DGRAD_1X1T_GFLOP", "100"))
# OPT-IN (measured no faster): the other implicit-GEMM data gradients (strided 3x3, 1x1 projections, the smaller head layers) over
# the transposed filter (conv_dgrad -> denet_conv_dgrad_t: the filter operand reduction-contiguous like the forward pass's,
# bit-identical results). Per launch, alone (tools/exp/per_launch.py): the strided 3x3 layers 352 -> 356 / 228 -> 224 / 202 -> 229 us,
# the head layers 517 -> 538 / 267 -> 306 us - what holds these launches back is not the filter's fragment reads (short reductions
# of 4-16 chunks per parity class, the strided gather of dy), so the default stays the k-major mode
DGRAD_T = os.environ.get("DENET_DGRAD_T", "0") != "0"
# the 3x3 stride-2 data gradients with the four parity classes of input pixels in one workgroup (csrc/dgrad_s2.hip); 0: the
# implicit-GEMM kernel, one problem per class
DGRAD_S2 = os.environ.get("DENET_DGRAD_S2", "1") != "0"


class wgrad_stream:
    """context: filter-gradient work of a layer (wgrad, bias column sums) on a second stream, ordered after everything
    queued so far on the compute stream. The data-gradient chain continues on the compute stream meanwhile: the
    tails of the many short kernels of one chain overlap with the heads of the other, and the HBM-bound transforms of
    one with the MFMA-bound products of the other. join_wgrad_stream() before the gradients are consumed."""

    def __enter__(self):
        global _WGRAD_STREAM, _ON_WGRAD_STREAM
        self.active = WGRAD_STREAM and (PROFILE is None or not PROFILE.alone)
        if not self.active:
            return self
        if _WGRAD_STREAM is None:
            init_streams()
        _WGRAD_STREAM.wait_stream(torch.cuda.current_stream())
        self.ctx = torch.cuda.stream(_WGRAD_STREAM)
        self.ctx.__enter__()
        _ON_WGRAD_STREAM = True
        return self

    def __exit__(self, *a):
        global _ON_WGRAD_STREAM
        if self.active:
            _ON_WGRAD_STREAM = False
            self.ctx.__exit__(*a)
        return False


def join_wgrad_stream():
    if _WGRAD_STREAM is not None:
        torch.cuda.current_stream().wait_stream(_WGRAD_STREAM)


def _wino_ws(tile, N, H, W, C, K):
    # the second stream has its own transform workspace (the two chains run concurrently)
    return WS.get("wino_side" if _ON_WGRAD_STREAM else "wino", _L().denet_conv_wino_workspace_bytes(tile, N, H, W, C, K))


def conv_wino_fwd(x, w, bias=None, add=None, out=None, tile=2, u=None, v_keep=None, relu=False, stats=None, up=None):
    """stats = (float64 buffer, cache dict): the output transform also writes the batch-norm column sums (conv_fwd bn_stats);
    up (UpLink, x = None; tiles 2 / 4 with stats only): the input is the 2 x 2 up-sampling of up.src, read in the input transform"""
    N, H, W, C = up.shape if up is not None else x.shape
    K = w.shape[0]
    y = out if out is not None else empty(N, H, W, K)
    if tile == FUSED2:
        import ctypes
        if u is None:
            u = conv_wino_filter(w, 2, dgrad=False)
        st, cache = stats if stats is not None and not relu else (None, None)
        rows = ctypes.c_int(0)
        with _armed(cache.get("bn_final") if st is not None else None) as fin:
            check(_L().denet_conv_wino2f(ptr(x), ptr(u), ptr(bias), ptr(add), ptr(y), int(relu), ptr(st), st.numel() * 8 if st is not None
                                         else 0, ctypes.byref(rows), N, H, W, C, K, stream_ptr()), "conv_wino2f")
        if st is not None:
            cache["bn_stats"] = _stats_result(st, rows.value, fin)
        return y
    if tile == FUSED4:
        import ctypes
        if u is None:
            u = conv_wino4t_filter(w, dgrad=False)
        st, cache = stats if stats is not None and not relu else (None, None)
        rows = ctypes.c_int(0)
        check(_L().denet_conv_wino4t_sums(ptr(x), ptr(u), ptr(bias), ptr(add), ptr(y), int(relu), ptr(st), st.numel() * 8 if st is not None
                                          else 0, ctypes.byref(rows), None, N, H, W, C, K, stream_ptr()), "conv_wino4t")
        if st is not None:
            cache["bn_stats"] = _stats_result(st, rows.value, None)
        return y
    ws = _wino_ws(tile, N, H, W, C, K)
    if stats is not None and not relu:
        import ctypes
        st, cache = stats
        rows = ctypes.c_int(0)
        fn = _L().denet_conv_wino_fwd_stats_up if up is not None else _L().denet_conv_wino_fwd_stats
        with _armed(cache.get("bn_final")) as fin:
            check(fn(ptr(up.src if up is not None else x), ptr(w), ptr(u), ptr(v_keep), ptr(bias), ptr(add), ptr(y), ptr(st),
                     st.numel() * 8, ctypes.byref(rows), ptr(ws), ws.numel(), tile, N, H, W, C, K, stream_ptr()), "conv_wino_fwd_stats")
        cache["bn_stats"] = _stats_result(st, rows.value, fin)
        return y
    assert up is None, "an up-sampled input is only read by the statistics form"
    check(_L().denet_conv_wino_fwd_act(ptr(x), ptr(w), ptr(u), ptr(v_keep), ptr(bias), ptr(add), ptr(y), int(relu), ptr(ws),
                                       ws.numel(), tile, N, H, W, C, K, stream_ptr()), "conv_wino_fwd")
    return y


def conv_wino_dgrad(dy, w, add=None, out=None, tile=2, u=None, sums=None, cache=None):
    """sums (BnSums): dx is the gradient of that batch norm's output - its backward reductions are written along (sums.partial)"""
    import ctypes
    N, H, W, K = dy.shape
    C = w.shape[3]
    dx = out if out is not None else empty(N, H, W, C)
    rows = ctypes.c_int(0)
    so = sums.c_struct() if sums is not None else None
    if tile == FUSED2:
        if u is None:
            u = conv_wino_filter(w, 2, dgrad=True)
        sb = sums.buffer(cache, N * ((H + 15) // 16) * ((W + 15) // 16), C) if sums is not None else None
        with _armed(sums.final if sums is not None else None) as fin:
            check(_L().denet_conv_wino2f_sums(ptr(dy), ptr(u), None, ptr(add), ptr(dx), 0, ptr(sb), sb.numel() * 8 if sb is not None else 0,
                                              ctypes.byref(rows), ctypes.byref(so) if so is not None else None, N, H, W, K, C,
                                              stream_ptr()), "conv_wino2f")
        if sums is not None:
            sums.done(sb, rows.value, fin)
        return dx
    if tile == FUSED4:
        if u is None:
            u = conv_wino4t_filter(w, dgrad=True)
        sb = sums.buffer(cache, _L().denet_conv_wino4t_stats_rows(N, H, W), C) if sums is not None else None
        check(_L().denet_conv_wino4t_sums(ptr(dy), ptr(u), None, ptr(add), ptr(dx), 0, ptr(sb), sb.numel() * 8 if sb is not None else 0,
                                          ctypes.byref(rows), ctypes.byref(so) if so is not None else None, N, H, W, K, C,
                                          stream_ptr()), "conv_wino4t")
        if sums is not None:
            sums.done(sb, rows.value, None)
        return dx
    ws = _wino_ws(tile, N, H, W, C, K)
    if sums is not None:
        T = N * -(-H // tile) * -(-W // tile)
        sb = sums.buffer(cache, (T * (C // 4) + 255) // 256, C)
        with _armed(sums.final) as fin:
            check(_L().denet_conv_wino_dgrad_sums(ptr(dy), ptr(w), ptr(u), ptr(add), ptr(dx), ctypes.byref(so), ptr(sb), sb.numel() * 8,
                                                  ctypes.byref(rows), ptr(ws), ws.numel(), tile, N, H, W, C, K, stream_ptr()),
                  "conv_wino_dgrad_sums")
        sums.done(sb, rows.value, fin)
        return dx
    check(_L().denet_conv_wino_dgrad(ptr(dy), ptr(w), ptr(u), ptr(add), ptr(dx), ptr(ws), ws.numel(), tile, N, H, W, C, K,
                                     stream_ptr()), "conv_wino_dgrad")
    return dx


def conv_wino_wgrad(x, dy, out=None, tile=2, v=None):
    N, H, W, C = x.shape
    K = dy.shape[3]
    dw = out if out is not None else empty(K, 3, 3, C)
    if tile == FUSED2:
        ws = WS.get("wino2f_side" if _ON_WGRAD_STREAM else "wino2f", _L().denet_conv_wino2f_wgrad_workspace_bytes(N, H, W))
        check(_L().denet_conv_wino2f_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(ws), ws.numel(), N, H, W, C, K, stream_ptr()),
              "conv_wino2f_wgrad")
        return dw
    ws = _wino_ws(tile, N, H, W, C, K)
    sws = WS.get("wgrad", WGRAD_WS_BYTES)
    check(_L().denet_conv_wino_wgrad(ptr(x), ptr(dy), ptr(v), ptr(dw), ptr(ws), ws.numel(), ptr(sws), sws.numel(), tile, N, H,
                                     W, C, K, stream_ptr()), "conv_wino_wgrad")
    return dw


def conv_wino_filter(w, tile, dgrad, out=None):
    """transformed filters [(tile+2)^2, K, C] (dgrad: [.., C, K]) for the u= argument of conv_wino_fwd / _dgrad"""
    K, _, _, C = w.shape
    nx = (tile + 2) * (tile + 2)
    u = out if out is not None else empty(nx, K, C)
    check(_L().denet_conv_wino_filter(ptr(w), ptr(u), tile, int(dgrad), C, K, stream_ptr()), "conv_wino_filter")
    return u


def conv_wino4t_filter(w, dgrad, out=None):
    """the F(4x4) transformed filters in the layout of the tile-parallel fused kernel, [red / 16][36][out][16] (red = the pass's
    reduction channels: C forward, K for the data gradient): denet_conv_wino_filter (tile 4) + denet_conv_wino4t_pack"""
    K, _, _, C = w.shape
    # (the prefetch runs on a side stream beside passes that may transform filters themselves: a scratch buffer per stream)
    u = WS.get("w4t_filter_%x" % torch.cuda.current_stream().cuda_stream, 36 * K * C * 4)[:36 * K * C * 4].view(torch.float32)
    conv_wino_filter(w, 4, dgrad, out=u)
    pk = out if out is not None else empty(36 * K * C)
    red, outc = (K, C) if dgrad else (C, K)
    check(_L().denet_conv_wino4t_pack(ptr(u), ptr(pk), red, outc, stream_ptr()), "conv_wino4t_pack")
    return pk


def conv_dgrad(dy, w, x_shape, add=None, stride=1, pad=0, s_real=None, out=None, logical=None, cache=None, sums=None):
    g = conv_geom(x_shape, w.shape, stride, pad, s_real)
    assert tuple(dy.shape) == (g[0], g[10], g[11], g[4]), (dy.shape, g)
    dx = out if out is not None else empty(*x_shape)
    if add is None and _bf16x3_geom(g):
        N, H, W, C, K = g[0], g[1], g[2], g[3], g[4]
        if cache is not None:
            cache["dgrad_tile"] = 0
        return _gemm_bf16x3(dy, _transpose(w, K, C), None, dx, N * H * W, C, K)       # dx[pix][c] = dy[pix][k] (w^T)[c][k]^T
    if DGRAD_1X1T_GFLOP > 0 and g[5] == 1 and g[6] == 1 and g[8] == 1 and g[9] == 0 and g[3] % 32 == 0 and g[4] % 32 == 0 \
            and 2e-9 * dy.numel() * g[3] >= DGRAD_1X1T_GFLOP:
        # a large 1x1 layer (the head): the forward kernel over the transposed filter (denet_conv_dgrad_1x1t, bit-identical);
        # the transposed copy comes from the side stream (wino_prefetch_filters) when a training step prepared it
        import ctypes
        N, H, W, C, K = g[0], g[1], g[2], g[3], g[4]
        wt = _cached_wt(cache)
        if wt is None:
            wt = _transpose(w, K, C)
        if cache is not None:
            cache["dgrad_tile"] = 0
            cache["dgrad_1x1t"] = True
        gt = conv_geom((N, H, W, K), (C, 1, 1, K), 1, 0, None)
        if not _tune_first(0, gt, dy, wt, None, add, dx, None) and PROFILE is not None:
            PROFILE.add(_conv_flops(g, logical))
        rows = ctypes.c_int(0)
        sb = so = None
        if sums is not None and (int(BWD_SUMS) & 2):
            sb = sums.buffer(cache, (N * H * W + 127) // 128, C)
            so = sums.c_struct()
        with _armed(sums.final if sb is not None else None) as fin:
            check(_L().denet_conv_dgrad_1x1t(ptr(dy), ptr(wt), ptr(add), ptr(dx), ctypes.byref(so) if so is not None else None, ptr(sb),
                                             sb.numel() * 8 if sb is not None else 0, ctypes.byref(rows), N, H, W, C, K, stream_ptr()),
                  "conv_dgrad_1x1t")
        if sb is not None:
            sums.done(sb, rows.value, fin)
        return dx
    if DGRAD_S2 and g[5] == 3 and g[6] == 3 and g[7] == 3 and g[8] == 2 and g[9] == 1 and \
            _L().denet_conv_dgrad_s2_ok(g[0], g[1], g[2], g[3], g[4]):
        # a 3x3 stride-2 layer (the first convolution of a ResNet stage): the four parity classes of input pixels in one
        # workgroup (csrc/dgrad_s2.hip); the packed filter comes from the side stream when a training step prepared it
        import ctypes
        N, H, W, C, K = g[0], g[1], g[2], g[3], g[4]
        wp = _cached_ws2(cache)
        if wp is None:
            wp = conv_dgrad_s2_pack(w)
        if cache is not None:
            cache["dgrad_tile"] = 0
            cache["dgrad_s2"] = True
        if PROFILE is not None:
            PROFILE.add(_conv_flops(g, logical))
        rows = ctypes.c_int(0)
        sb = so = None
        if sums is not None and (int(BWD_SUMS) & 2):
            sb = sums.buffer(cache, _L().denet_conv_dgrad_s2_stats_rows(N, H, W), C)
            so = sums.c_struct()
        check(_L().denet_conv_dgrad_s2(ptr(dy), ptr(wp), ptr(add), ptr(dx), ctypes.byref(so) if so is not None else None, ptr(sb),
                                       sb.numel() * 8 if sb is not None else 0, ctypes.byref(rows), N, H, W, C, K, stream_ptr()),
              "conv_dgrad_s2")
        if sb is not None:
            sums.done(sb, rows.value, None)
        return dx
    _tune_first(1, g, dy, w, None, add, dx, None)

    def direct():
        check(_L().denet_conv_dgrad(ptr(dy), ptr(w), ptr(add), ptr(dx), *g, stream_ptr()), "conv_dgrad")

    tile = _wino_tile(1, g, direct, lambda t: conv_wino_dgrad(dy, w, add, out=dx, tile=t))
    if tile:
        if PROFILE is not None:
            PROFILE.add(_conv_flops(g, logical) / _WINO_GAIN[tile])
        u = None
        if cache is not None:
            cache["dgrad_tile"] = tile
            u = _cached_u(cache, 1, tile)
        return conv_wino_dgrad(dy, w, add, out=dx, tile=tile, u=u, sums=sums if (int(BWD_SUMS) & 1) else None, cache=cache)
    if cache is not None:
        cache["dgrad_tile"] = 0
    if PROFILE is not None:
        PROFILE.add(_conv_flops(g, logical))
    if DGRAD_T and g[3] % 32 == 0 and g[4] % 32 == 0:
        # the implicit-GEMM data gradient over the TRANSPOSED filter wt [R][S][C][K] (denet_conv_dgrad_t: the filter operand
        # reduction-contiguous like the forward pass's; same products in the same order, bit-identical). The transposed copy comes
        # from the side stream (wino_prefetch_filters) when a training step prepared it
        import ctypes
        N, H, W, C, K = g[0], g[1], g[2], g[3], g[4]
        wt = _cached_wt(cache)
        if wt is None:
            wt = _transpose(w, K, w.numel() // K)
        if cache is not None:
            cache["dgrad_t"] = True
        rows = ctypes.c_int(0)
        sb = so = None
        if sums is not None and (int(BWD_SUMS) & 2) and g[1] % g[8] == 0 and g[2] % g[8] == 0:
            sb = sums.buffer(cache, (N * H * W + 127) // 128 + g[8] * g[8], C)
            so = sums.c_struct()
        with _armed(sums.final if sb is not None else None) as fin:
            check(_L().denet_conv_dgrad_t(ptr(dy), ptr(wt), ptr(add), ptr(dx), ctypes.byref(so) if so is not None else None, ptr(sb),
                                          sb.numel() * 8 if sb is not None else 0, ctypes.byref(rows), *g, stream_ptr()), "conv_dgrad_t")
        if sb is not None:
            sums.done(sb, rows.value, fin)
        return dx
    if sums is not None and (int(BWD_SUMS) & 2) and g[1] % g[8] == 0 and g[2] % g[8] == 0:
        # the epilogue of the implicit-GEMM kernel leaves the batch norm's backward reductions behind as well (a row of sums per
        # row tile, and per parity class of input pixels when the layer strides)
        import ctypes
        N, H, W, C = g[0], g[1], g[2], g[3]
        rows = ctypes.c_int(0)
        sb = sums.buffer(cache, (N * H * W + 127) // 128 + g[8] * g[8], C)
        so = sums.c_struct()
        with _armed(sums.final) as fin:
            check(_L().denet_conv_dgrad_sums(ptr(dy), ptr(w), ptr(add), ptr(dx), ctypes.byref(so), ptr(sb), sb.numel() * 8,
                                             ctypes.byref(rows), *g, stream_ptr()), "conv_dgrad_sums")
        sums.done(sb, rows.value, fin)
        return dx
    direct()
    return dx


def conv_wgrad(x, dy, w_shape, stride=1, pad=0, s_real=None, out=None, logical=None, cache=None):
    g = conv_geom(x.shape, w_shape, stride, pad, s_real)
    assert tuple(dy.shape) == (g[0], g[10], g[11], g[4]), (dy.shape, g)
    dw = out if out is not None else empty(*w_shape)
    if _bf16x3_geom(g):
        N, H, W, C, K = g[0], g[1], g[2], g[3], g[4]
        M = N * H * W                                                                   # dw[k][c] = sum over pixels dy^T[k][pix] x^T[c][pix]
        return _gemm_bf16x3(_transpose(dy, M, K), _transpose(x, M, C), None, dw, K, C, M)
    ws = WS.get("wgrad", WGRAD_WS_BYTES)
    _tune_first(2, g, x, dy, None, None, dw, ws)

    def direct():
        check(_L().denet_conv_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(ws), ws.numel(), *g, stream_ptr()), "conv_wgrad")

    tile = _wino_tile(2, g, direct, lambda t: conv_wino_wgrad(x, dy, out=dw, tile=t))
    if tile:
        if PROFILE is not None:
            PROFILE.add(_conv_flops(g, logical) / _WINO_GAIN[tile])
        v = None
        if cache is not None and cache.get("V_tile") == tile:
            v = cache.get("V")
            cache["V_tile"] = None            # consumed: the next forward pass writes a fresh one
        return conv_wino_wgrad(x, dy, out=dw, tile=tile, v=v)
    if PROFILE is not None:
        PROFILE.add(_conv_flops(g, logical))
    direct()
    return dw


UP_LINK = os.environ.get("DENET_UP_LINK", "1") != "0"


class UpLink:
    """the output of a pool-inverse layer (2 x 2 nearest-neighbour up-sampling, pool_inv.py:10-41) that has not been written: a
    Winograd convolution behind it reads the small tensor inside its input transform (denet_conv_wino_fwd_stats_up), anybody else
    calls materialise(); result = the up-sampled tensor once it exists"""

    def __init__(self, src):
        self.src = src
        N, H, W, C = src.shape
        self.shape = (N, 2 * H, 2 * W, C)
        self.result = None

    def materialise(self):
        if self.result is None:
            self.result = pool_inv_fwd(self.src, 2, 2)
        return self.result


class NchwLink:
    """the network input as the reference feeds it, [N,3,H,W] (dataset/__init__.py:359): the first layer's own kernels read it in
    that layout (csrc/stem.hip), the NHWC tensor every other consumer expects is only made when somebody asks for it"""

    def __init__(self, x, cp):
        self.x, self.cp = x, cp

    def materialise(self):
        return nchw_to_nhwc(self.x, self.cp)


def conv_stem_ok(x_nchw, w_shape, stride, pad, s_real):
    """the first layer's kernels take this convolution straight from the planar image batch"""
    N, C, H, W = x_nchw.shape
    K, R, S, cp = w_shape
    a = (N, H, W, cp, K, R, S, s_real, stride, pad, H // 2, W // 2)
    return C == 3 and bool(_L().denet_conv_stem_ok(0, *a)) and bool(_L().denet_conv_stem_ok(1, *a))


def conv_stem_fwd(x_nchw, w, bias, cache, bn_stats, logical=None, relu=False):
    """y = conv7x7/2(x) + bias from the planar image batch (+ the batch-norm column sums: cache["bn_stats"], see conv_fwd; or,
    relu, the ReLU of the inference fold)"""
    import ctypes
    N, _, H, W = x_nchw.shape
    K = w.shape[0]
    g = conv_geom((N, H, W, w.shape[3]), tuple(w.shape), 2, 3, 7)
    y = empty(N, H // 2, W // 2, K)
    st, rows = None, ctypes.c_int(0)
    if bn_stats:
        st = cache.get("bn_stats_buf")
        if st is None or st.numel() < 1024 * 2 * K:          # a row per workgroup: at most one per CU
            st = cache["bn_stats_buf"] = torch.empty(1024 * 2 * K, dtype=torch.float64, device="cuda")
    if PROFILE is not None:
        PROFILE.add(_conv_flops(g, logical))
    check(_L().denet_conv_stem_fwd_act(ptr(x_nchw), 1, ptr(w), ptr(bias), ptr(y), int(bool(relu)), ptr(st),
                                       st.numel() * 8 if st is not None else 0, ctypes.byref(rows), N, H, W, stream_ptr()), "conv_stem_fwd")
    cache["bn_stats"] = (st, rows.value) if st is not None else None
    return y


def conv_stem_wgrad(x_nchw, dy, w_shape, out, logical=None):
    N, _, H, W = x_nchw.shape
    g = conv_geom((N, H, W, w_shape[3]), tuple(w_shape), 2, 3, 7)
    ws = WS.get("wgrad", WGRAD_WS_BYTES)
    if PROFILE is not None:
        PROFILE.add(_conv_flops(g, logical))
    check(_L().denet_conv_stem_wgrad_from(ptr(x_nchw), 1, ptr(dy), ptr(out), ptr(ws), ws.numel(), N, H, W, stream_ptr()),
          "conv_stem_wgrad")
    return out


def _bn_ws(M, C):
    return WS.get("bn", _L().denet_bn_workspace_bytes(M, C))


def bn_fwd_train(x, gamma, beta, run_mean, run_stdinv, momentum=0.9, eps=1e-5, relu=False, res=None, out=None, pre=None):
    """pre = (partial sums float64 [rows][2][C], rows) written by the convolution that produced x (conv_fwd bn_stats): the
    statistics pass over x is skipped"""
    C = x.shape[-1]
    M = x.numel() // C
    y = out if out is not None else torch.empty_like(x)
    save_mean, save_invstd = empty(C), empty(C)
    if pre is not None and len(pre) > 2:
        # the convolution that produced x has finished the statistics itself (BnFinal): only the pointwise pass is left
        fin = pre[2]
        check(_L().denet_bn_apply(ptr(x), ptr(res), ptr(y), ptr(gamma), ptr(beta), ptr(fin.save_mean), ptr(fin.save_invstd), M, C,
                                  int(relu), stream_ptr()), "bn_apply")
        return y, fin.save_mean, fin.save_invstd
    if pre is not None:
        check(_L().denet_bn_fwd_train_pre(ptr(x), ptr(res), ptr(y), ptr(gamma), ptr(beta), ptr(run_mean), ptr(run_stdinv),
                                          ptr(save_mean), ptr(save_invstd), ptr(pre[0]), int(pre[1]), M, C, momentum, eps,
                                          int(relu), stream_ptr()), "bn_fwd_train_pre")
        return y, save_mean, save_invstd
    check(_L().denet_bn_fwd_train(ptr(x), ptr(res), ptr(y), ptr(gamma), ptr(beta), ptr(run_mean), ptr(run_stdinv),
                                  ptr(save_mean), ptr(save_invstd), ptr(_bn_ws(M, C)), M, C, momentum, eps,
                                  int(relu), stream_ptr()), "bn_fwd_train")
    return y, save_mean, save_invstd


# ---- batch norm whose pointwise pass is left to the consumer (include/denet_hip.h: denet_bn_link) -------------------------------
# Measured (round 3): the first form of the transform kernel (one thread per tile gathering the halo through L2 for two or three
# tensors) was 15-40 % slower than the passes it replaced; with the patch staged through LDS and the loads of a batch of rounds
# issued together it is faster: 965 -> 979 img/s for the whole step. DENET_BN_LINK=0 restores the separate passes.
LINK_BN = os.environ.get("DENET_BN_LINK", "1") != "0"


LINK_COUNT = [0, 0]        # forward / backward passes that took the linked transform (tests)
SUMS_COUNT = [0]           # data-gradient passes that also produced a batch norm's backward sums (tests)
# the output transform of a Winograd data-gradient pass also writes the two backward reductions of the batch norm whose output
# gradient it produces (denet_conv_wino_dgrad_sums / denet_conv_wino2f_sums): that layer's backward then needs no pass of its own
# over (gradient, input, output) for them. The sums are accumulated per block in fp32 before they are widened (as the forward
# statistics from the convolution epilogues are), so they differ from bn_bwd_partial_kernel's in the last bits. 0: off
BWD_SUMS = int(os.environ.get("DENET_BN_BWD_SUMS", "3"))      # bit 0: Winograd passes, bit 1: direct passes (1x1 head, strided 3x3 layers)


def _bn_link_struct(x, aux, y, gamma, beta, mean, invstd, coef, out, relu):
    import ctypes

    class _C(ctypes.Structure):
        _fields_ = [(n, ctypes.c_void_p) for n in ("x", "aux", "y", "gamma", "beta", "mean", "invstd", "coef", "out")] + \
                   [("relu", ctypes.c_int)]
    return _C(ptr(x), ptr(aux), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(coef), ptr(out), int(relu))


class BnSums:
    """request to a data-gradient pass: `the tensor you write is the gradient of the output of this batch norm - leave its two
    backward reductions behind` (x = the layer's input, y = its forward output if the ReLU mask needs it); partial = (float64
    buffer [rows][2][C], rows) afterwards, or None when the pass that ran cannot"""

    def __init__(self, x, y, gamma, beta, mean, invstd, relu, final=None):
        self.x, self.y, self.gamma, self.beta, self.mean, self.invstd, self.relu = x, y, gamma, beta, mean, invstd, bool(relu)
        self.partial = None
        self.final = final          # BnFinal (kind 2): the pass may also finish the reduction (dgamma, dbeta, coef); None: never

    def c_struct(self):
        return _bn_link_struct(self.x, None, self.y, self.gamma, self.beta, self.mean, self.invstd, None, None, self.relu)

    def buffer(self, cache, rows, C):
        buf = cache.get("bsum_buf") if cache is not None else None
        if buf is None or buf.numel() < rows * 2 * C:
            buf = torch.empty(rows * 2 * C, dtype=torch.float64, device="cuda")
            if cache is not None:
                cache["bsum_buf"] = buf
        return buf

    def done(self, buf, rows, fin=None):
        self.partial = None
        if rows > 0:
            self.partial = (buf, rows, fin) if (fin is not None and fin.taken) else (buf, rows)
            SUMS_COUNT[0] += 1


class BnLink:
    """A batch-norm layer that has reduced its statistics (forward) / its two gradient sums (backward) and leaves the pointwise
    pass to whoever reads the tensor next: a Winograd convolution evaluates it inside its input transform
    (denet_conv_wino_fwd_fold / denet_conv_wino_dgrad_fold), anybody else calls materialise(), which launches the very kernel
    the batch norm would have launched (denet_bn_apply / denet_bn_bwd_apply). Values are bit-identical either way."""

    def __init__(self, backward, x, aux, y, gamma, beta, mean, invstd, coef, relu, out=None):
        self.backward, self.x, self.aux, self.y = backward, x, aux, y
        self.gamma, self.beta, self.mean, self.invstd, self.coef, self.relu = gamma, beta, mean, invstd, coef, bool(relu)
        self.out = out            # backward: the residual-branch gradient buffer (dres) or None; forward: set by the taker
        self.result = None        # the tensor the pointwise pass produces (forward: activation, backward: dx), once it exists

    def c_struct(self, out):
        return _bn_link_struct(self.x, self.aux, self.y, self.gamma, self.beta, self.mean, self.invstd, self.coef, out, self.relu)

    def materialise(self):
        if self.result is not None:
            return self.result
        C = self.x.shape[-1]
        M = self.x.numel() // C
        r = torch.empty_like(self.x)
        if not self.backward:
            check(_L().denet_bn_apply(ptr(self.x), ptr(self.aux), ptr(r), ptr(self.gamma), ptr(self.beta), ptr(self.mean),
                                      ptr(self.invstd), M, C, int(self.relu), stream_ptr()), "bn_apply")
        else:
            check(_L().denet_bn_bwd_apply(ptr(self.x), ptr(self.y), ptr(self.aux), ptr(self.gamma), ptr(self.beta), ptr(self.mean),
                                          ptr(self.invstd), ptr(self.coef), ptr(r), ptr(self.out), M, C, int(self.relu),
                                          stream_ptr()), "bn_bwd_apply")
        self.result = r
        return r


def bn_fwd_train_link(x, gamma, beta, run_mean, run_stdinv, pre, momentum=0.9, eps=1e-5, relu=False, res=None):
    """bn_fwd_train(pre=...) without the pointwise pass: coefficients + running statistics now, the activation when somebody
    reads it (BnLink). Returns (link, save_mean, save_invstd)."""
    C = x.shape[-1]
    M = x.numel() // C
    if len(pre) > 2:           # the producing convolution has finished the statistics itself (BnFinal)
        save_mean, save_invstd = pre[2].save_mean, pre[2].save_invstd
    else:
        save_mean, save_invstd = empty(C), empty(C)
        check(_L().denet_bn_stats_final(ptr(pre[0]), int(pre[1]), M, C, momentum, eps, ptr(run_mean), ptr(run_stdinv), ptr(save_mean),
                                        ptr(save_invstd), stream_ptr()), "bn_stats_final")
    return BnLink(False, x, res, None, gamma, beta, save_mean, save_invstd, None, relu), save_mean, save_invstd


def bn_bwd_link(x, y, dy, gamma, save_mean, save_invstd, relu=False, want_dres=False, dgamma=None, dbeta=None, beta=None,
                pre=None):
    """bn_bwd without the pointwise pass: dgamma / dbeta / the two means now, dx (and dres) when somebody reads them. Returns
    (link, dres buffer or None). pre = (partial sums float64 [rows][2][C], rows) left behind by the data-gradient pass that
    produced dy (BnSums): the reduction pass over dy, x and y is skipped."""
    C = x.shape[-1]
    M = x.numel() // C
    dgamma = dgamma if dgamma is not None else empty(C)
    dbeta = dbeta if dbeta is not None else empty(C)
    coef = None
    if pre is not None and len(pre) > 2 and pre[2].dgamma.data_ptr() == dgamma.data_ptr() and pre[2].dbeta.data_ptr() == dbeta.data_ptr():
        coef = pre[2].coef      # the data-gradient pass that wrote dy has finished the two sums itself (BnFinal)
    if coef is not None:
        pass
    elif pre is not None:
        coef = empty(2 * C)
        check(_L().denet_bn_bwd_final(ptr(pre[0]), int(pre[1]), M, C, ptr(dgamma), ptr(dbeta), ptr(coef), stream_ptr()), "bn_bwd_final")
    else:
        coef = empty(2 * C)
        check(_L().denet_bn_bwd_sums(ptr(x), ptr(y), ptr(dy), ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_invstd), ptr(dgamma),
                                     ptr(dbeta), ptr(coef), ptr(_bn_ws(M, C)), M, C, int(relu), stream_ptr()), "bn_bwd_sums")
    dres = torch.empty_like(x) if want_dres else None
    return BnLink(True, x, dy, y, gamma, beta, save_mean, save_invstd, coef, relu, out=dres), dres


WEIGHTS_VERSION = 0       # bumped whenever parameters / running statistics change (solver step, set_value, packing)


def bump_weights_version():
    """invalidates what inference derives from the weights once and keeps: BN test coefficients, transformed filters"""
    global WEIGHTS_VERSION
    WEIGHTS_VERSION += 1


INFER_FOLD = os.environ.get("DENET_INFER_FOLD", "1") != "0"      # inference: batch norm folded into the convolution in front


def bn_fold(w, conv_bias, gamma, beta, run_mean, run_stdinv, eps):
    """filters / bias of the convolution that equals conv(x, w) + conv_bias followed by test-mode batch norm"""
    K = w.shape[0]
    w_f, b_f = torch.empty_like(w), empty(K)
    check(_L().denet_bn_fold(ptr(w), ptr(conv_bias), ptr(gamma), ptr(beta), ptr(run_mean), ptr(run_stdinv), float(eps),
                             ptr(w_f), ptr(b_f), K, w.numel() // K, stream_ptr()), "bn_fold")
    return w_f, b_f


def bn_fwd_test(x, gamma, beta, run_mean, run_stdinv, eps=1e-5, relu=False, res=None, out=None, cache=None):
    """cache: a dict owned by the layer; its inference coefficients are computed once per weights version"""
    C = x.shape[-1]
    M = x.numel() // C
    y = out if out is not None else torch.empty_like(x)
    ent = cache.get("test_coef") if cache is not None else None
    ready = ent is not None and ent[1] == WEIGHTS_VERSION
    coef = ent[0] if ent is not None else empty(2 * C)
    check(_L().denet_bn_fwd_test(ptr(x), ptr(res), ptr(y), ptr(gamma), ptr(beta), ptr(run_mean), ptr(run_stdinv),
                                 ptr(coef), int(ready), M, C, eps, int(relu), stream_ptr()), "bn_fwd_test")
    if cache is not None and not ready:
        cache["test_coef"] = (coef, WEIGHTS_VERSION)
    return y


def bn_bwd(x, y, dy, gamma, save_mean, save_invstd, relu=False, want_dres=False, dgamma=None, dbeta=None, dx=None,
           beta=None):
    """y may be None for a relu layer without residual when beta is given: the mask is recomputed from x"""
    C = x.shape[-1]
    M = x.numel() // C
    dx = dx if dx is not None else torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    dgamma = dgamma if dgamma is not None else empty(C)
    dbeta = dbeta if dbeta is not None else empty(C)
    check(_L().denet_bn_bwd(ptr(x), ptr(y), ptr(dy), ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_invstd), ptr(dx), ptr(dres),
                            ptr(dgamma), ptr(dbeta), ptr(_bn_ws(M, C)), M, C, int(relu), stream_ptr()), "bn_bwd")
    return dx, dres, dgamma, dbeta


BN_POOL_FUSE = os.environ.get("DENET_BN_POOL_FUSE", "1") != "0"    # training: BN + ReLU + max pool without the tensor in between


def bn_relu_pool_fwd_train(x, gamma, beta, run_mean, run_stdinv, k, stride, pad, momentum=0.9, eps=1e-5, pre=None, xhat=False):
    """relu(bn(x)) max-pooled, without writing relu(bn(x)): returns (y_pool, argmax, save_mean, save_invstd); values and
    argmax taps are those of bn_fwd_train(relu=True) + maxpool_fwd. xhat: also (x - mean) * invstd at each window's argmax, as a
    fifth result - what bn_relu_pool_bwd_sums (the backward reductions over the pooled tensors) reads"""
    N, H, W, C = x.shape
    OH = (H + 2 * pad - k) // stride + 1
    OW = (W + 2 * pad - k) // stride + 1
    y = empty(N, OH, OW, C)
    arg = torch.empty((N, OH, OW, C), dtype=torch.uint8, device="cuda")
    xh = empty(N, OH, OW, C) if xhat else None
    save_mean, save_invstd = empty(C), empty(C)
    ws = _bn_ws(N * H * W, C)
    check(_L().denet_bn_relu_pool_fwd_train_xhat(ptr(x), ptr(y), ptr(arg), ptr(xh), ptr(gamma), ptr(beta), ptr(run_mean),
                                                 ptr(run_stdinv), ptr(save_mean), ptr(save_invstd),
                                                 ptr(pre[0]) if pre is not None else None, int(pre[1]) if pre is not None else 0,
                                                 ptr(ws), N, H, W, C, OH, OW, k, stride, pad, momentum, eps, stream_ptr()),
          "bn_relu_pool_fwd_train")
    return (y, arg, save_mean, save_invstd, xh) if xhat else (y, arg, save_mean, save_invstd)


_CONST_VEC = {}


def const_vec(value, C):
    """[C] floats of one value on the device (mean 0 / invstd 1 of an already normalised tensor)"""
    v = _CONST_VEC.get((value, C))
    if v is None:
        v = _CONST_VEC[(value, C)] = torch.full((C,), float(value), dtype=torch.float32, device="cuda")
    return v


def bn_relu_pool_bwd_pooled(x, xhat_pool, y_pool, dy_pool, arg, gamma, beta, save_mean, save_invstd, k, stride, pad, dgamma, dbeta,
                            pre=None):
    """gradient of bn_relu_pool_fwd_train(xhat=True) with the two reductions taken over the POOLED tensors (every window sends its
    gradient to its argmax pixel, whose ReLU output is y_pool): a quarter of the elements and no window gather. pre = (partial sums,
    rows) left by the data-gradient pass that wrote dy_pool (BnSums on the pooled tensors): no reduction pass at all. Same sums as
    bn_relu_pool_bwd up to the order of a double-precision summation. Returns dx."""
    N, H, W, C = x.shape
    OH, OW = dy_pool.shape[1], dy_pool.shape[2]
    coef = empty(2 * C)
    if pre is not None:
        check(_L().denet_bn_bwd_final(ptr(pre[0]), int(pre[1]), N * H * W, C, ptr(dgamma), ptr(dbeta), ptr(coef), stream_ptr()),
              "bn_bwd_final")
    else:
        check(_L().denet_bn_relu_pool_bwd_sums(ptr(xhat_pool), ptr(y_pool), ptr(dy_pool), ptr(const_vec(0.0, C)), ptr(const_vec(1.0, C)),
                                               ptr(dgamma), ptr(dbeta), ptr(coef), ptr(_bn_ws(N * OH * OW, C)), N, H, W, C, OH, OW,
                                               stream_ptr()), "bn_relu_pool_bwd_sums")
    dx = torch.empty_like(x)
    check(_L().denet_bn_relu_pool_bwd_apply(ptr(x), ptr(dy_pool), ptr(arg), ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_invstd),
                                            ptr(coef), ptr(dx), N, H, W, C, OH, OW, k, stride, pad, stream_ptr()),
          "bn_relu_pool_bwd_apply")
    return dx


def bn_relu_pool_bwd(x, dy_pool, arg, gamma, beta, save_mean, save_invstd, k, stride, pad, dgamma=None, dbeta=None):
    """gradient of bn_relu_pool_fwd_train: (dx, dgamma, dbeta); bit-identical to maxpool_bwd + bn_bwd(relu=True)"""
    N, H, W, C = x.shape
    OH, OW = dy_pool.shape[1], dy_pool.shape[2]
    dx = torch.empty_like(x)
    dgamma = dgamma if dgamma is not None else empty(C)
    dbeta = dbeta if dbeta is not None else empty(C)
    check(_L().denet_bn_relu_pool_bwd(ptr(x), ptr(dy_pool), ptr(arg), ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_invstd),
                                      ptr(dx), ptr(dgamma), ptr(dbeta), ptr(_bn_ws(N * H * W, C)), N, H, W, C, OH, OW, k, stride,
                                      pad, stream_ptr()), "bn_relu_pool_bwd")
    return dx, dgamma, dbeta


def maxpool_fwd(x, k, stride, pad):
    N, H, W, C = x.shape
    OH = (H + 2 * pad - k) // stride + 1
    OW = (W + 2 * pad - k) // stride + 1
    y = empty(N, OH, OW, C)
    arg = torch.empty((N, OH, OW, C), dtype=torch.uint8, device="cuda")
    check(_L().denet_maxpool_fwd(ptr(x), ptr(y), ptr(arg), N, H, W, C, OH, OW, k, stride, pad, stream_ptr()),
          "maxpool_fwd")
    return y, arg


def maxpool_bwd(dy, arg, x_shape, k, stride, pad):
    N, H, W, C = x_shape
    OH, OW = dy.shape[1], dy.shape[2]
    dx = empty(*x_shape)
    check(_L().denet_maxpool_bwd(ptr(dy), ptr(arg), ptr(dx), N, H, W, C, OH, OW, k, stride, pad, stream_ptr()),
          "maxpool_bwd")
    return dx


def avgpool_fwd(x, k, stride, pad):
    N, H, W, C = x.shape
    OH = (H + 2 * pad - k) // stride + 1
    OW = (W + 2 * pad - k) // stride + 1
    y = empty(N, OH, OW, C)
    check(_L().denet_avgpool_fwd(ptr(x), ptr(y), N, H, W, C, OH, OW, k, stride, pad, stream_ptr()), "avgpool_fwd")
    return y


def avgpool_bwd(dy, x_shape, k, stride, pad):
    N, H, W, C = x_shape
    dx = empty(*x_shape)
    check(_L().denet_avgpool_bwd(ptr(dy), ptr(dx), N, H, W, C, dy.shape[1], dy.shape[2], k, stride, pad,
                                 stream_ptr()), "avgpool_bwd")
    return dx


def pool_inv_fwd(x, fy, fx):
    N, H, W, C = x.shape
    y = empty(N, H * fy, W * fx, C)
    check(_L().denet_pool_inv_fwd(ptr(x), ptr(y), N, H, W, C, fy, fx, stream_ptr()), "pool_inv_fwd")
    return y


def pool_inv_bwd(dy, fy, fx):
    N, OH, OW, C = dy.shape
    dx = empty(N, OH // fy, OW // fx, C)
    check(_L().denet_pool_inv_bwd(ptr(dy), ptr(dx), N, OH // fy, OW // fx, C, fy, fx, stream_ptr()), "pool_inv_bwd")
    return dx


# ---- shape / stochastic layers (csrc/augment.hip) -----------------------------------------------------------
def border_fwd(x, border):
    """border = (left, right, top, bottom) (denet/layer/border.py:18)"""
    N, H, W, C = x.shape
    l, r, t, b = (int(v) for v in border)
    y = empty(N, H + t + b, W + l + r, C)
    check(_L().denet_border_fwd(ptr(x), ptr(y), N, H, W, C, l, r, t, b, stream_ptr()), "border_fwd")
    return y


def border_bwd(dy, border):
    N, OH, OW, C = dy.shape
    l, r, t, b = (int(v) for v in border)
    dx = empty(N, OH - t - b, OW - l - r, C)
    check(_L().denet_border_bwd(ptr(dy), ptr(dx), N, OH - t - b, OW - l - r, C, l, r, t, b, stream_ptr()), "border_bwd")
    return dx


def crop_mirror_fwd(x, crop, mirror_pr, flip_pr, train, seed):
    """crop = (rows, cols); seed: 64-bit counter key of this (layer, iteration)"""
    N, H, W, C = x.shape
    y = empty(N, int(crop[0]), int(crop[1]), C)
    check(_L().denet_crop_mirror_fwd(ptr(x), ptr(y), N, H, W, C, int(crop[0]), int(crop[1]), float(mirror_pr),
                                     float(flip_pr), int(bool(train)), int(seed), stream_ptr()), "crop_mirror_fwd")
    return y


def crop_mirror_bwd(dy, x_shape, mirror_pr, flip_pr, train, seed):
    N, H, W, C = x_shape
    dx = empty(N, H, W, C)
    check(_L().denet_crop_mirror_bwd(ptr(dy), ptr(dx), N, H, W, C, dy.shape[1], dy.shape[2], float(mirror_pr),
                                     float(flip_pr), int(bool(train)), int(seed), stream_ptr()), "crop_mirror_bwd")
    return dx


def dropout(x, c_logical, rate, seed):
    """y = x * mask(seed) / (1 - rate); apply to dy with the same seed for the gradient"""
    N, C = x.shape[0], x.shape[-1]
    y = torch.empty_like(x)
    check(_L().denet_dropout(ptr(x), ptr(y), N, x.numel() // (N * C), C, int(c_logical), float(rate), int(seed),
                             stream_ptr()), "dropout")
    return y


def concat_fwd(a, b, ca, cb, cyp):
    """logical channel concatenation of two channel-padded NHWC buffers"""
    rows = a.numel() // a.shape[-1]
    y = empty(*a.shape[:-1], cyp)
    check(_L().denet_concat_fwd(ptr(a), ptr(b), ptr(y), rows, int(ca), a.shape[-1], int(cb), b.shape[-1], int(cyp),
                                stream_ptr()), "concat_fwd")
    return y


def concat_bwd(dy, ca, cap, cb, cbp):
    rows = dy.numel() // dy.shape[-1]
    da = empty(*dy.shape[:-1], cap)
    db = empty(*dy.shape[:-1], cbp)
    check(_L().denet_concat_bwd(ptr(dy), ptr(da), ptr(db), rows, int(ca), int(cap), int(cb), int(cbp), dy.shape[-1],
                                stream_ptr()), "concat_bwd")
    return da, db


def add_bias(x, bias, out=None):
    y = out if out is not None else torch.empty_like(x)
    C = x.shape[-1]
    check(_L().denet_add_bias(ptr(x), ptr(bias), ptr(y), x.numel() // C, C, stream_ptr()), "add_bias")
    return y


def layer_seed(base, layer_index, iteration):
    """64-bit counter key of one stochastic layer at one training iteration"""
    return (int(base) * 0x9E3779B97F4A7C15 + (int(layer_index) + 1) * 0xD1B54A32D192ED03
            + (int(iteration) + 1) * 0x8CB92BA72F3D8DD7) & 0xFFFFFFFFFFFFFFFF


def nchw_to_nhwc(x, cp):
    N, C, H, W = x.shape
    y = empty(N, H, W, cp)
    check(_L().denet_nchw_to_nhwc(ptr(x), ptr(y), N, C, H, W, cp, stream_ptr()), "nchw_to_nhwc")
    return y


def nhwc_to_nchw(x, c):
    N, H, W, CP = x.shape
    y = empty(N, c, H, W)
    check(_L().denet_nhwc_to_nchw(ptr(x), ptr(y), N, c, H, W, CP, stream_ptr()), "nhwc_to_nchw")
    return y


def add(a, b, relu=False, out=None):
    y = out if out is not None else torch.empty_like(a)
    check(_L().denet_add(ptr(a), ptr(b), ptr(y), a.numel(), int(relu), stream_ptr()), "add")
    return y


def relu_fwd(x):
    y = torch.empty_like(x)
    check(_L().denet_relu_fwd(ptr(x), ptr(y), x.numel(), stream_ptr()), "relu_fwd")
    return y


def relu_bwd(y, dy):
    dx = torch.empty_like(dy)
    check(_L().denet_relu_bwd(ptr(y), ptr(dy), ptr(dx), y.numel(), stream_ptr()), "relu_bwd")
    return dx


def colsum(x, out=None):
    C = x.shape[-1]
    M = x.numel() // C
    out = out if out is not None else empty(C)
    ws = WS.get("colsum_side" if _ON_WGRAD_STREAM else "colsum", _L().denet_colsum_workspace_bytes(M, C))
    check(_L().denet_colsum(ptr(x), ptr(out), ptr(ws), M, C, stream_ptr()), "colsum")
    return out


def solver_step(params, moments, grads, n_decay, lr, momentum, iteration, decay, mode, grad_scale=1.0):
    check(_L().denet_solver_step(ptr(params), ptr(moments), ptr(grads), params.numel(), int(n_decay), lr, momentum,
                                 int(iteration), decay, grad_scale, int(mode), stream_ptr()), "solver_step")


def solver_adam(params, m, v, grads, n_decay, lr, beta1, beta2, iteration, decay, grad_scale=1.0):
    check(_L().denet_solver_adam(ptr(params), ptr(m), ptr(v), ptr(grads), params.numel(), int(n_decay), lr, beta1, beta2,
                                 int(iteration), decay, grad_scale, stream_ptr()), "solver_adam")


def corner_fwd(conv, cn):
    B, H, W, CP = conv.shape
    pr = empty(B, 2, cn, H, W)
    check(_L().denet_corner_fwd(ptr(conv), ptr(pr), B, H, W, CP, cn, stream_ptr()), "corner_fwd")
    return pr


def _loss_ws():
    return WS.get("loss", _L().denet_loss_workspace_bytes())


def corner_loss(corner_pr, target, dconv, cost_out, cost_factor):
    B, _, cn, H, W = corner_pr.shape
    CP = dconv.shape[-1] if dconv is not None else 0
    check(_L().denet_corner_loss(ptr(corner_pr), ptr(target), ptr(dconv), ptr(cost_out), ptr(_loss_ws()), B, H, W, CP,
                                 cn, cost_factor, stream_ptr()), "corner_loss")


def sparse_fwd_buffers(M, gs, kp):
    """the two outputs of sparse_fwd for M RoIs, allocated ahead of the call (the training step makes them before it waits for the RoI
    proposal: nothing but the launch stands between the bbox array and the gather)"""
    return empty(M, kp), torch.empty((M, gs * gs), dtype=torch.int32, device="cuda")


def sparse_fwd(fmap, bbox, coff, F, rois_per_image, gs, kp, tap_rule=0, buffers=None):
    B, H, W, CP = fmap.shape
    M = B * rois_per_image
    out, taps = buffers if buffers is not None else sparse_fwd_buffers(M, gs, kp)
    assert tuple(out.shape) == (M, kp) and tuple(taps.shape) == (M, gs * gs)
    check(_L().denet_sparse_fwd(ptr(fmap), ptr(bbox), ptr(out), ptr(taps), B, H, W, CP, coff, F, rois_per_image, gs,
                                kp, tap_rule, stream_ptr()), "sparse_fwd")
    return out, taps


_SORT_STREAM = None


def sparse_sort_async(taps, B, H, W, rois_per_image, gs):
    """queues the tap sort of the gather gradient on a side stream right after the forward gather (it needs only the
    taps); returns the event sparse_bwd(presorted=...) waits for"""
    global _SORT_STREAM
    if _SORT_STREAM is None:
        init_streams()
    ws = WS.get("sparse_sort", _L().denet_sparse_sort_workspace_bytes(B, H, W, rois_per_image, gs))
    if bool(_L().denet_sparse_sort_is_single(B, H, W, rois_per_image, gs)):
        # the one-kernel sort (one 1024-thread workgroup per image, 128 KB of LDS): on a side stream its workgroups starve for
        # LDS beside the head's matrix kernels until those drain (1.5-1.9 ms in the trace); alone it takes ~50 us, so it runs
        # here, on the compute stream, right behind the gather
        check(_L().denet_sparse_sort(ptr(taps), ptr(ws), ws.numel(), B, H, W, rois_per_image, gs, stream_ptr()), "sparse_sort")
        ev = torch.cuda.Event()
        ev.record()
        return ev
    _SORT_STREAM.wait_stream(torch.cuda.current_stream())          # taps are written by sparse_fwd on this stream
    with torch.cuda.stream(_SORT_STREAM):
        check(_L().denet_sparse_sort(ptr(taps), ptr(ws), ws.numel(), B, H, W, rois_per_image, gs, stream_ptr()), "sparse_sort")
        ev = torch.cuda.Event()
        ev.record(_SORT_STREAM)
    return ev


def sparse_bwd(dy, taps, dfmap, coff, F, rois_per_image, gs, zero_from, presorted=None):
    B, H, W, CP = dfmap.shape
    kp = dy.shape[-1]
    ws = WS.get("sparse_sort", _L().denet_sparse_sort_workspace_bytes(B, H, W, rois_per_image, gs))
    if presorted is not None:
        torch.cuda.current_stream().wait_event(presorted)
    check(_L().denet_sparse_bwd(ptr(dy), None if presorted is not None else ptr(taps), ptr(ws), ws.numel(), ptr(dfmap), B, H, W, CP,
                                coff, F, rois_per_image, gs, kp, zero_from, stream_ptr()), "sparse_bwd")
    return dfmap


def detect_loss(logits, det_target, bbox_valid, bbox_target, roi_bbox, dlogits, costs, batch, ncls, nreg,
                cost_factor, bbox_factor, bounded_iou=False, fit_target=None, nfit=0, fit_factor=0.0):
    M, CP = logits.shape
    check(_L().denet_detect_loss(ptr(logits), ptr(det_target), ptr(bbox_valid), ptr(bbox_target), ptr(roi_bbox),
                                 ptr(fit_target), ptr(dlogits), ptr(costs), ptr(_loss_ws()), M, batch, CP, ncls, nreg,
                                 int(nfit), cost_factor, bbox_factor, float(fit_factor), int(bounded_iou), stream_ptr()),
          "detect_loss")


def build_samples(corner_pr, corner_threshold, sample_count, max_corners=1024, local_max=0, out=None):
    """Device part of build_samples: returns (box int32 [B,S,4], absd fp32 [B,S], count int32 [B]) on device."""
    B, _, cn, H, W = corner_pr.shape
    nbytes = _L().denet_build_samples_workspace_bytes(B, cn, H, W, max_corners, sample_count)
    ws = WS.get("samples", nbytes)
    if out is not None:
        box, absd, count = out
    else:
        box = torch.empty((B, sample_count, 4), dtype=torch.int32, device="cuda")
        absd = empty(B, sample_count)
        count = torch.empty((B,), dtype=torch.int32, device="cuda")
    check(_L().denet_build_samples(ptr(corner_pr), ptr(box), ptr(absd), ptr(count), ptr(ws), ws.numel(), B, cn, H, W,
                                   corner_threshold, sample_count, max_corners, local_max, stream_ptr()),
          "build_samples")
    return box, absd, count


def build_samples_stats(corner_pr, sample_count, max_corners=1024):
    """diagnostics of the last build_samples call of this geometry: (corners kept [B, Cn] int32, pair candidates [B] int64),
    host tensors (synchronises)"""
    B, _, cn, H, W = corner_pr.shape
    nbytes = _L().denet_build_samples_workspace_bytes(B, cn, H, W, max_corners, sample_count)
    ws = WS.get("samples", nbytes)
    nc = torch.empty((B, cn), dtype=torch.int32, device="cuda")
    cand = torch.empty((B,), dtype=torch.int32, device="cuda")
    check(_L().denet_build_samples_stats(ptr(ws), ws.numel(), B, cn, H, W, max_corners, sample_count, ptr(nc), ptr(cand),
                                         stream_ptr()), "build_samples_stats")
    return nc.cpu(), cand.cpu().to(torch.int64) & 0xFFFFFFFF


def samples_finish_host(box, absd, count, H, W, out=None):
    """Host epilogue (libm expf, double box arithmetic exactly as the reference). CPU tensors in, CPU tensor out (`out`: a
    buffer of the caller's to write into - a fresh 0.4 MB allocation costs ~0.1 ms of page faults inside the RoI hand-off)."""
    B, S, _ = box.shape
    if out is None or tuple(out.shape) != (B, S, 5):
        out = torch.empty((B, S, 5), dtype=torch.float32)
    check(_L().denet_samples_finish_host(box.data_ptr(), absd.data_ptr(), count.data_ptr(), B, S, H, W,
                                         out.data_ptr()), "samples_finish_host")
    return out


def cluster_samples_host(raw, counts, threshold, output_num):
    """apply_cluster per image (denet_sparse.cc:541-542): raw [B, K, 5] ranked candidates, counts [B]; images with more than
    output_num candidates are clustered. Returns (numpy [B, output_num, 5], numpy counts [B])"""
    import ctypes
    import numpy
    raw = numpy.ascontiguousarray(raw, dtype=numpy.float32)
    B = raw.shape[0]
    out = numpy.zeros((B, output_num, 5), dtype=numpy.float32)
    out_counts = numpy.zeros(B, dtype=numpy.int32)
    n_out = ctypes.c_int(0)
    for b in range(B):
        n = int(counts[b])
        if n > output_num:
            check(_L().denet_host_cluster_samples(raw[b].ctypes.data, n, float(threshold), int(output_num), out[b].ctypes.data,
                                                  ctypes.byref(n_out)), "cluster_samples")
            out_counts[b] = n_out.value
        else:
            out[b, :n] = raw[b, :n]
            out_counts[b] = n
    return out, out_counts


def detect_decode(logits, roi_bbox, class_num, jointfit, nreg, overlap_threshold, nfit=0):
    M, CP = logits.shape
    det_pr, fitness, bbox = empty(M, class_num + 1), empty(M, class_num + 1), empty(M, 4)
    check(_L().denet_detect_decode(ptr(logits), ptr(roi_bbox), ptr(det_pr), ptr(fitness), ptr(bbox), M, CP, class_num,
                                   int(jointfit), nreg, int(nfit), float(overlap_threshold), stream_ptr()), "detect_decode")
    return det_pr, fitness, bbox


def detect_nms(det_pr, fitness, bbox, count, B, S, class_num, pr_threshold, nms_threshold):
    keep = torch.empty((B, class_num, S), dtype=torch.uint8, device="cuda")
    check(_L().denet_detect_nms(ptr(det_pr), ptr(fitness), ptr(bbox), ptr(count), ptr(keep), B, S, class_num,
                                float(pr_threshold), float(nms_threshold), stream_ptr()), "detect_nms")
    return keep


def soft_nms_batch_host(det_h, fit_h, box_h, counts, B, S, class_num, pr_threshold, nms_threshold):
    """the soft-NMS tail of a whole batch in one native call -> (log-domain scores fp32, classes, rows = b*S + RoI, per-image counts)"""
    import numpy
    cap = int(counts.sum()) * class_num + 1
    score = numpy.empty(cap, dtype=numpy.float32)
    cls = numpy.empty(cap, dtype=numpy.int32)
    row = numpy.empty(cap, dtype=numpy.int32)
    per = numpy.zeros(B, dtype=numpy.int32)
    counts = numpy.ascontiguousarray(counts, dtype=numpy.int32)
    n = _L().denet_soft_nms_batch_host(det_h.ctypes.data, fit_h.ctypes.data, box_h.ctypes.data, counts.ctypes.data, B, S,
                                       class_num, float(pr_threshold), float(nms_threshold), score.ctypes.data,
                                       cls.ctypes.data, row.ctypes.data, per.ctypes.data, cap)
    if n < 0:
        check(int(n), "soft_nms_batch_host")
    return score[:n], cls[:n], row[:n], per


def soft_nms_batch(det_pr, fitness, bbox, count, B, S, class_num, pr_threshold, nms_threshold):
    """the soft-NMS tail of a whole batch on the DEVICE (one wave per (class, image)) -> numpy (log-domain scores fp32, classes,
    rows = b*S + RoI, per-image counts): what soft_nms_batch_host returns, bit for bit. Two device-to-host copies: the total
    (with the per-image counts), then the `total` entries."""
    import numpy
    P = B * class_num
    ws_bytes = int(_L().denet_soft_nms_workspace_bytes(B, S, class_num))
    # workspace and outputs from the grow-only pool (118 MB at B=32, S=2304, 80 classes: not through the caching allocator on
    # every call of the inference loop)
    ws = WS.get("soft_nms", ws_bytes)
    score = WS.get("soft_nms_score", P * S * 4).view(torch.float32)[:P * S]
    cr = WS.get("soft_nms_cr", 2 * P * S * 4).view(torch.int32)[:2 * P * S].view(2, P * S)
    head = WS.get("soft_nms_head", (B + 1) * 4).view(torch.int32)[:B + 1]       # per-image counts, then the total
    check(_L().denet_soft_nms_batch(ptr(det_pr), ptr(fitness), ptr(bbox), ptr(count), B, S, class_num, float(pr_threshold),
                                    float(nms_threshold), ptr(score), ptr(cr[0]), ptr(cr[1]), ptr(head),
                                    ptr(head) + 4 * B, ptr(ws), ws_bytes, stream_ptr()), "soft_nms_batch")
    head_h = head.cpu().numpy()
    n = int(head_h[B])
    cr_h = cr[:, :n].cpu().numpy()
    return score[:n].cpu().numpy(), cr_h[0], cr_h[1], head_h[:B].copy()


def soft_nms_host(score, box, nms_threshold):
    """numpy in / numpy out: (order, final scores) of one class' candidates (Gaussian soft-NMS)"""
    import ctypes
    import numpy
    n = int(score.shape[0])
    score = numpy.ascontiguousarray(score, dtype=numpy.float32)
    box = numpy.ascontiguousarray(box, dtype=numpy.float32)
    order = numpy.empty(max(n, 1), dtype=numpy.int32)
    out = numpy.empty(max(n, 1), dtype=numpy.float32)
    k = ctypes.c_int(0)
    check(_L().denet_soft_nms_host(score.ctypes.data, box.ctypes.data, n, float(nms_threshold), order.ctypes.data,
                                   out.ctypes.data, ctypes.addressof(k)), "soft_nms_host")
    return order[:k.value], out[:k.value]
