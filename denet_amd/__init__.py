"""denet_amd — MI355X-native implementation of the DeNet training hot path."""
import os

_tuned = False


def host_tuning():
    """Host-side runtime settings of the training process (called once when a model is packed onto the device).

    * torch intra-op threads -> 1: the hot path uses torch for storage/streams only, but its OpenMP pool
      (one thread per core, spin-waiting) burns the container's CPU quota and gets the launch thread throttled
      for tens of milliseconds (measured on the GPU box: 16-core quota, 256 hardware threads).
    * glibc malloc: serve the step's numpy temporaries (0.1-6 MB) from the heap instead of mmap/munmap — with the
      GPU driver attached every munmap runs MMU notifiers, which made a 0.4 ms host phase take 5.6 ms.
    """
    global _tuned
    if _tuned:
        return
    _tuned = True
    try:
        import torch
        if "DENET_TORCH_THREADS" in os.environ:
            torch.set_num_threads(int(os.environ["DENET_TORCH_THREADS"]))
        else:
            torch.set_num_threads(1)
    except Exception:
        pass
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        M_TRIM_THRESHOLD, M_MMAP_THRESHOLD = -1, -3
        libc.mallopt(M_MMAP_THRESHOLD, 1 << 30)
        libc.mallopt(M_TRIM_THRESHOLD, 1 << 30)
    except Exception:
        pass
