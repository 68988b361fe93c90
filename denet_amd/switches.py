"""Every DENET_* environment switch of the build in ONE place: name -> (product default, what it selects).

The product default is the environment with NONE of them set; they exist for A / B measurements and for the tests that pin an
alternative path. bench.py prints the ones that are set (config.denet_switches) and whether the run was the product default;
tests/test_host.py checks that every switch the sources read is listed here. "kernels": the switch changes which kernels a training
step launches (a bench line taken with it set is not the headline configuration)."""

SWITCHES = {
    # ---- which implementation a convolution pass uses ------------------------------------------------------------------
    "DENET_AUTOTUNE": ("1", "kernels", "use the committed decisions and the Winograd / fused algorithms (0: launch heuristics, direct kernels only)"),
    "DENET_TUNE": ("0", "kernels", "1: MEASURE the launch configuration / algorithm of geometries the tuned file does not cover on their first call (what tools/tune.py does); 0, the product default: such a geometry runs ops.static_policy, nothing is ever timed, every process runs the same kernels"),
    "DENET_TUNE_CACHE": ("denet_amd/tuned/gfx950.json", "kernels", "the committed measured decisions (0: ignore the file)"),
    "DENET_WINOGRAD": ("4", "kernels", "largest Winograd tile allowed for the 3x3 stride-1 layers (0: direct only, 2: F(2x2) only)"),
    "DENET_WINO_RAGGED": ("1", "kernels", "Winograd on maps that are no multiple of the tile (ceil tiles; 0: multiples only)"),
    "DENET_WINO2F": ("7", "kernels", "fused F(2x2) kernels of the 64-channel layers: bit 0 forward, 1 data gradient, 2 filter gradient"),
    "DENET_DGRAD_S2": ("1", "kernels", "3x3 stride-2 data gradients: the four parity classes in one workgroup (csrc/dgrad_s2.hip; 0: implicit GEMM per class)"),
    "DENET_WINO4T": ("3", "kernels", "tile-parallel fully fused F(4x4) kernel (csrc/wino4t.hip) where the tuned file names it: bit 0 forward, 1 data gradient"),
    "DENET_W4T_LDS": ("0", "kernels", "experiment: LDS bytes a wino4t workgroup requests (100000: one workgroup per CU)"),
    "DENET_WINO4F": ("1", "kernels", "fused F(4x4) product + output-transform kernel (0: un-fused component GEMMs + transform)"),
    "DENET_WINO4F_TB": ("0", "kernels", "force a shape of the fused F(4x4) kernel (32 / 33 / 34 / 64; 0: the occupancy policy)"),
    "DENET_WINO4G": ("1", "kernels", "dedicated F(4x4) filter-gradient kernel (0: the generic batched split-K products)"),
    "DENET_STEM": ("3", "kernels", "the first layer's own kernels: bit 0 forward, bit 1 filter gradient (0: generic implicit GEMM)"),
    "DENET_HEAD_BF16X3": ("0", "kernels", "OPT-IN, never the headline: head GEMMs as 3-term bf16 splits on the bf16 matrix cores"),
    "DENET_DGRAD_FIRST_GFLOP": ("100", "kernels", "a layer whose data-gradient GEMM is at least this large runs it before its filter gradient is queued"),
    "DENET_DGRAD_1X1T_GFLOP": ("100", "kernels", "1x1 stride-1 data gradients at least this large run as forward products over the transposed filter (0: never)"),
    "DENET_DGRAD_T": ("0", "kernels", "OPT-IN (no faster): the other implicit-GEMM data gradients over the transposed filter (igemm mode 3)"),
    "DENET_DGRAD_T_NBUF": ("0", "kernels", "experiment: loop structure of igemm mode 3"),
    "DENET_DGRAD_T_TILE": ("0", "kernels", "experiment: tile of igemm mode 3"),
    "DENET_IGEMM_NBUF": ("0", "kernels", "experiment: force the LDS buffering of the implicit-GEMM kernels"),
    "DENET_IGEMM_TILE": ("0", "kernels", "experiment: force the implicit-GEMM tile"),
    "DENET_WGRAD_BLOCKS": ("0", "kernels", "experiment: force the split-K workgroup count of the filter gradient"),
    "DENET_WGRAD_SPLIT_SLOW": ("1", "kernels", "filter gradient: the split slice is the slow (XCD-local) workgroup index"),
    # ---- batch norm and its neighbours ---------------------------------------------------------------------------------
    "DENET_BN_LINK": ("1", "kernels", "batch-norm pointwise passes evaluated inside the Winograd transforms next to them"),
    "DENET_BN_BWD_SUMS": ("3", "kernels", "backward reductions written by the data-gradient pass: bit 0 Winograd passes, bit 1 direct passes"),
    "DENET_BN_FINAL_FOLD": ("0", "kernels", "OPT-IN (slower): the reductions' second stage inside the producing launch: bit 0 forward, bit 1 backward"),
    "DENET_BN_POOL_FUSE": ("1", "kernels", "the stem's BN + ReLU + max pool as one pass each way"),
    "DENET_BN_ACT_FUSE": ("1", "kernels", "`BN A` layer pairs run as the fused BN + ReLU"),
    "DENET_POOL_QUAD": ("1", "kernels", "the stem's backward pointwise pass on 2x2 pixel quads"),
    "DENET_PREP_LDS": ("1", "kernels", "the linked transforms stage their block through LDS (0: one thread per tile)"),
    "DENET_SKIP_FUSE": ("1", "kernels", "SKIP additions in the epilogue of the convolution in front"),
    "DENET_UP_LINK": ("1", "kernels", "the pool-inverse layer read inside the next convolution's input transform"),
    "DENET_INFER_FOLD": ("1", "kernels", "inference: batch norm folded into the convolution in front of it"),
    # ---- streams, RoI path ---------------------------------------------------------------------------------------------
    "DENET_WGRAD_STREAM": ("1", "kernels", "the filter-gradient chain of the backward sweep on a second stream"),
    "DENET_SHORT_HANDOFF": ("1", "host", "the short forms of the RoI hand-off (device-side editing / one native call)"),
    "DENET_HANDOFF_WARM_MS": ("0.25", "host", "milliseconds between dry runs of the hand-off's native call while the host waits for the proposal (0: none)"),
    "DENET_SIDE_SORT": ("1", "kernels", "the gather gradient's tap sort queued right behind the forward gather"),
    "DENET_SORT_ONE_KERNEL": ("1", "kernels", "the tap sort as one 1024-thread workgroup per image"),
    "DENET_SOFT_NMS_HOST": ("unset", "host", "inference: force the host (1) / device (0) form of Gaussian soft-NMS"),
    # ---- drivers / tooling ---------------------------------------------------------------------------------------------
    "DENET_FORCE_DP": ("unset", "driver", "run the data-parallel collectives at world size 1 (single-GPU exercise of the RCCL path)"),
    "DENET_BENCH_SHARE_GPU": ("unset", "driver", "bench.py --share-gpu: all ranks on cuda:0 over gloo (launch-path debug mode)"),
    "DENET_BUILD_JOBS": ("cpu count", "build", "parallel hipcc processes of denet_amd/build.py"),
    "DENET_TORCH_THREADS": ("unset", "driver", "torch CPU thread count set when the package is imported"),
    "DENET_TEST_PRESSURE": ("unset", "tests", "every GPU test starts beside a saturated memory system (tests/conftest.py)"),
}


def active():
    """{name: value} of the switches set in this process's environment"""
    import os
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("DENET_")}


def changes_kernels(env=None):
    """the set switches that change which kernels a step launches (bench.py: product_default_switches is `not changes_kernels()`);
    an unknown DENET_* name counts"""
    env = active() if env is None else env
    return [k for k in env if SWITCHES.get(k, (None, "kernels"))[1] in ("kernels", "host")]
