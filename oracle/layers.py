"""ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under denet_amd/ may import this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.

CPU (numpy) restatement of the arithmetic of the reference's hot-path layers, NCHW float32 like the reference.
Every function cites the reference lines it follows (paths relative to /root/reference).

PARITY PIN: none of the reference's Python layers can be imported here (every module imports theano at the
top; Theano is not installed) and the third-party arithmetic they lower to (Theano conv2d, cuDNN BN / pooling)
is not in the tree, so this restatement is pinned only by
  * the reference's own BN known-answer block (denet/layer/batch_norm.py:131-154) -> test_oracle.py,
  * the build_samples known answers recorded in SURVEY.md §8(c) -> tests/golden/build_samples_kat.json,
  * agreement with an independent second implementation (torch CPU ops + autograd) to 1e-5 -> test_oracle.py.
Everything else is "parity unpinned" (see DESIGN.md).
"""
import math

import numpy as np

F32 = np.float32


# ---------------------------------------------------------------------------------------------------------
# convolution: denet/layer/convolution.py:55-89. Theano conv2d = TRUE convolution (filters flipped),
# border 'half' = pad k//2, 'valid' = 0, 'full' = k-1, int n = n; output = ceil((in + 2p - k + 1)/s).
# ---------------------------------------------------------------------------------------------------------
_SCRATCH = {}
_SCRATCH_CAP = []
_POOL = []


def _par(n, fn, min_bytes=0, nbytes=0):
    """fn(lo, hi) over the images [0, n) in contiguous chunks on a thread pool (numpy releases the GIL inside array operations).
    Every chunk computes exactly what the serial expression computes for its images - element-wise work and per-image copies
    only, never a reduction across images: the results are bit-identical to the serial ones. At the benchmark sizes the oracle
    spent most of its time in single-threaded element-wise passes over gigabyte tensors (tools/exp/oracle_profile.py)."""
    import os
    if n < 2 or nbytes < (8 << 20):
        fn(0, n)
        return
    if not _POOL:
        from concurrent.futures import ThreadPoolExecutor
        _POOL.append(ThreadPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 2) // 2))))
    workers = _POOL[0]._max_workers
    step = max(1, -(-n // workers))
    futs = [_POOL[0].submit(fn, lo, min(n, lo + step)) for lo in range(0, n, step)]
    for f in futs:
        f.result()


def _scratch_cap():
    """bytes the reused work arrays may hold together: a quarter of the machine's memory, at most 24 GB"""
    if not _SCRATCH_CAP:
        try:
            import psutil
            total = psutil.virtual_memory().total
        except Exception:
            total = 16 << 30
        _SCRATCH_CAP.append(min(24 << 30, total // 4))
    return _SCRATCH_CAP[0]


def _scratch(tag, shape, dtype):
    """a reused work array (tag, shape, dtype): the im2col matrices of a 512x512 batch are gigabytes, and a fresh allocation of
    that size is page-faulted in on every call - most of the oracle's time at the benchmark sizes. Only for temporaries that do
    not outlive the call that asks for them."""
    key = (tag, tuple(shape), np.dtype(dtype).str)
    buf = _SCRATCH.get(key)
    if buf is None:
        if sum(b.nbytes for b in _SCRATCH.values()) + int(np.prod(shape)) * np.dtype(dtype).itemsize > _scratch_cap():
            _SCRATCH.clear()
        buf = _SCRATCH[key] = np.empty(shape, dtype=dtype)
    return buf


def _im2col(x, R, S, stride, pad):
    N, C, H, W = x.shape
    OH = (H + 2 * pad - R) // stride + 1
    OW = (W + 2 * pad - S) // stride + 1
    if pad > 0:
        xp = _scratch("pad", (N, C, H + 2 * pad, W + 2 * pad), x.dtype)
        xp[:, :, :pad] = 0
        xp[:, :, pad + H:] = 0
        xp[:, :, :, :pad] = 0
        xp[:, :, :, pad + W:] = 0
        xp[:, :, pad:pad + H, pad:pad + W] = x
    else:
        xp = x
    cols = _scratch("cols", (N, C, R, S, OH, OW), x.dtype)

    c2 = cols.reshape(N * C, R, S, OH, OW)             # (a view: the work array is contiguous)
    x2 = xp.reshape(N * C, xp.shape[2], xp.shape[3])   # images x channels as ONE axis: a single image still spreads over the pool

    def fill(lo, hi):
        for r in range(R):
            for s in range(S):
                c2[lo:hi, r, s] = x2[lo:hi, r:r + stride * OH:stride, s:s + stride * OW:stride]

    _par(N * C, fill, nbytes=cols.nbytes)
    return cols.reshape(N, C * R * S, OH * OW), OH, OW


def _col2im(cols, x_shape, R, S, stride, pad, OH, OW):
    N, C, H, W = x_shape
    xp = np.zeros((N, C, H + 2 * pad, W + 2 * pad), dtype=cols.dtype)
    cols = cols.reshape(N, C, R, S, OH, OW)

    c2 = cols.reshape(N * C, R, S, OH, OW)
    x2 = xp.reshape(N * C, H + 2 * pad, W + 2 * pad)

    def fold(lo, hi):
        for r in range(R):
            for s in range(S):
                x2[lo:hi, r:r + stride * OH:stride, s:s + stride * OW:stride] += c2[lo:hi, r, s]

    _par(N * C, fold, nbytes=cols.nbytes)
    return xp[:, :, pad:pad + H, pad:pad + W] if pad > 0 else xp


def conv2d(x, w, b=None, stride=1, pad=0):
    """y[n,o,i,j] = sum_{c,u,v} w[o,c,u,v] * x[n,c,i*s+(R-1-u)-p, j*s+(S-1-v)-p] (+ b[o])"""
    K, C, R, S = w.shape
    cols, OH, OW = _im2col(x, R, S, stride, pad)
    wf = w[:, :, ::-1, ::-1].reshape(K, C * R * S)
    y = np.matmul(wf[None], cols).reshape(x.shape[0], K, OH, OW)
    if b is not None:
        y = y + b[None, :, None, None]
    return y.astype(x.dtype)


def conv2d_grad(x, w, dy, stride=1, pad=0, need_dx=True):
    """gradients of conv2d w.r.t. x, w, b (what tensor.grad yields, denet/model/model_cnn.py:318)"""
    K, C, R, S = w.shape
    N = x.shape[0]
    cols, OH, OW = _im2col(x, R, S, stride, pad)
    dy2 = dy.reshape(N, K, OH * OW)
    dwf = np.einsum("nkp,ncp->kc", dy2, cols, optimize=True).reshape(K, C, R, S)
    dw = dwf[:, :, ::-1, ::-1]
    db = dy.sum(axis=(0, 2, 3))
    dx = None
    if need_dx:
        wf = w[:, :, ::-1, ::-1].reshape(K, C * R * S)
        dcols = np.matmul(wf.T[None], dy2, out=_scratch("dcols", (N, C * R * S, OH * OW), np.result_type(wf, dy2)))
        dx = _col2im(dcols, x.shape, R, S, stride, pad, OH, OW)
    return dx, np.ascontiguousarray(dw), db


def deconv2d(x, w, b=None, stride=1, pad=0):
    """`DC`: denet/layer/deconvolution.py:54-67. w is omega (C_out, C_in, kh, kw); the output is the gradient of the
    true convolution F (filters omega with the first two axes swapped, i.e. (C_in, C_out, kh, kw)) w.r.t. its input
    of shape (N, C_out, h, w), h = H*s - 2p + k - 1, evaluated at output-gradient x"""
    O, I, R, S = w.shape
    N, C, H, W = x.shape
    assert C == I
    h = H * stride - 2 * pad + R - 1
    wd = W * stride - 2 * pad + S - 1
    wt = np.ascontiguousarray(w.transpose(1, 0, 2, 3))          # F's filters (K = C_in, C = C_out)
    wf = wt[:, :, ::-1, ::-1].reshape(I, O * R * S)
    dcols = np.matmul(wf.T[None], x.reshape(N, I, H * W))
    y = _col2im(dcols, (N, O, h, wd), R, S, stride, pad, H, W)
    if b is not None:
        y = y + b[None, :, None, None]
    return y.astype(x.dtype)


def deconv2d_grad(x, w, dy, stride=1, pad=0, need_dx=True):
    """gradients of deconv2d w.r.t. x, omega, bias: dx = F(dy); d_omega = (filter gradient of F with input dy and
    output-gradient x), axes swapped back"""
    wt = np.ascontiguousarray(w.transpose(1, 0, 2, 3))
    dx = conv2d(dy, wt, None, stride, pad) if need_dx else None
    _, dwt, _ = conv2d_grad(dy, wt, x, stride, pad, need_dx=False)
    return dx, np.ascontiguousarray(dwt.transpose(1, 0, 2, 3)), dy.sum(axis=(0, 2, 3))


# ---------------------------------------------------------------------------------------------------------
# `B` / `D` / `CM` (denet/layer/border.py:18-33, dropout.py:20-24, crop_mirror.py:26-56). The reference's random
# stream is Theano's MRG_RandomStreams (third party, not under /root/reference): parity with it is UNPINNED. The
# build defines its masks by a counter-based generator (csrc/augment.hip); this is its restatement.
# ---------------------------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def mix64(seed, idx):
    """splitmix64 finaliser of seed + idx * golden; idx: uint64 array"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M64) + np.asarray(idx, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def u24(seed, idx):
    return (mix64(seed, idx) >> np.uint64(40)).astype(np.int64)


def thr24(p):
    return 0 if p <= 0.0 else (1 << 24) if p >= 1.0 else int(p * 16777216.0)


def layer_seed(base, layer_index, iteration):
    return (int(base) * 0x9E3779B97F4A7C15 + (int(layer_index) + 1) * 0xD1B54A32D192ED03
            + (int(iteration) + 1) * 0x8CB92BA72F3D8DD7) & _M64


def dropout_mask(shape, rate, seed):
    """mask * 1/(1-rate) over a logical NCHW shape; rate is rounded to fp32 like the C-ABI argument"""
    rate = float(np.float32(rate))
    keep = u24(seed, np.arange(int(np.prod(shape)), dtype=np.uint64)) < thr24(1.0 - rate)
    return (keep.astype(F32) * F32(1.0 / (1.0 - rate))).reshape(shape)


def crop_mirror_geom(N, H, W, CH, CW, mirror_pr, flip_pr, train, seed):
    """per-image (off_r, off_c, flip, mirror)"""
    if not train:
        return [((H - CH) // 2, (W - CW) // 2, False, False)] * N
    n4 = np.arange(N, dtype=np.uint64) * np.uint64(4)
    mirror = u24(seed, n4) > thr24(1.0 - float(np.float32(mirror_pr)))
    flip = u24(seed, n4 + np.uint64(1)) > thr24(1.0 - float(np.float32(flip_pr)))
    off_r = (u24(seed, n4 + np.uint64(2)) * (H - CH + 1)) >> 24
    off_c = (u24(seed, n4 + np.uint64(3)) * (W - CW + 1)) >> 24
    return [(int(off_r[n]), int(off_c[n]), bool(flip[n]), bool(mirror[n])) for n in range(N)]


def crop_mirror(x, crop, geom):
    N, C, H, W = x.shape
    y = np.empty((N, C, crop[0], crop[1]), dtype=x.dtype)
    for n, (r0, c0, flip, mirror) in enumerate(geom):
        v = x[n, :, r0:r0 + crop[0], c0:c0 + crop[1]]
        if flip:
            v = v[:, ::-1, :]
        if mirror:
            v = v[:, :, ::-1]
        y[n] = v
    return y


def crop_mirror_grad(dy, x_shape, geom):
    dx = np.zeros(x_shape, dtype=dy.dtype)
    CH, CW = dy.shape[2], dy.shape[3]
    for n, (r0, c0, flip, mirror) in enumerate(geom):
        v = dy[n]
        if flip:
            v = v[:, ::-1, :]
        if mirror:
            v = v[:, :, ::-1]
        dx[n, :, r0:r0 + CH, c0:c0 + CW] = v
    return dx


def border(x, b):
    """b = (left, right, top, bottom)"""
    return np.pad(x, ((0, 0), (0, 0), (b[2], b[3]), (b[0], b[1])))


def border_grad(dy, b):
    H, W = dy.shape[2], dy.shape[3]
    return np.ascontiguousarray(dy[:, :, b[2]:H - b[3], b[0]:W - b[1]])


# ---------------------------------------------------------------------------------------------------------
# batch normalisation: denet/layer/batch_norm.py:50-79 (cuDNN spatial BN: biased variance, eps inside the
# sqrt; running mean / running INVERSE std with momentum; test path feeds var=(1/stdinv)^2 and cuDNN adds eps
# again), denet/layer/batch_norm_relu.py:34-54 (ReLU (x+|x|)/2 after BN; grad masks dy by xn > 0).
# ---------------------------------------------------------------------------------------------------------
def bn_train(x, gamma, beta, eps=1e-5):
    xd = x.astype(np.float64)
    mean = xd.mean(axis=(0, 2, 3))
    var = xd.var(axis=(0, 2, 3))
    invstd = 1.0 / np.sqrt(var + eps)
    y = np.empty(x.shape, dtype=F32)

    N, C = x.shape[0], x.shape[1]
    rows = lambda v: np.tile(np.asarray(v), N)[:, None, None]           # per-channel values along the merged (image, channel) axis
    x2, y2 = xd.reshape(N * C, x.shape[2], x.shape[3]), y.reshape(N * C, x.shape[2], x.shape[3])
    m2, i2, g2, b2 = rows(mean), rows(invstd), rows(gamma), rows(beta)

    def norm(lo, hi):
        y2[lo:hi] = ((x2[lo:hi] - m2[lo:hi]) * i2[lo:hi] * g2[lo:hi] + b2[lo:hi]).astype(F32)

    _par(N * C, norm, nbytes=xd.nbytes)
    return y, mean.astype(F32), invstd.astype(F32)


def bn_running_update(run_mean, run_stdinv, mean, invstd, momentum=0.9):
    """batch_norm.py:75-76"""
    return (F32(momentum) * run_mean + F32(1.0 - momentum) * mean).astype(F32), \
           (F32(momentum) * run_stdinv + F32(1.0 - momentum) * invstd).astype(F32)


def bn_test(x, gamma, beta, run_mean, run_stdinv, eps=1e-5):
    """batch_norm.py:50-52: var = sqr(1/stdinv); dnn_batch_normalization_test(..., var, eps) -> double eps"""
    var = (F32(1.0) / run_stdinv) ** 2
    inv = 1.0 / np.sqrt(var.astype(np.float64) + eps)
    y = (x.astype(np.float64) - run_mean[None, :, None, None]) * inv[None, :, None, None] * gamma[None, :, None, None] \
        + beta[None, :, None, None]
    return y.astype(F32)


def bn_grad(x, dy, gamma, mean, invstd):
    """cuDNN BN backward with saved mean / invstd: returns dx, dgamma, dbeta"""
    xd, dyd = x.astype(np.float64), dy.astype(np.float64)
    m = x.shape[0] * x.shape[2] * x.shape[3]
    N, C, H, W = x.shape
    rows = lambda v: np.tile(np.asarray(v), N)[:, None, None]           # per-channel values along the merged (image, channel) axis
    xh = np.empty(xd.shape, dtype=np.float64)
    prod = np.empty(xd.shape, dtype=np.float64)
    x2, d2, h2, p2 = xd.reshape(N * C, H, W), dyd.reshape(N * C, H, W), xh.reshape(N * C, H, W), prod.reshape(N * C, H, W)
    m2, i2 = rows(mean), rows(invstd)

    def hat(lo, hi):
        h2[lo:hi] = (x2[lo:hi] - m2[lo:hi]) * i2[lo:hi]
        p2[lo:hi] = d2[lo:hi] * h2[lo:hi]

    _par(N * C, hat, nbytes=xd.nbytes)
    dbeta = dyd.sum(axis=(0, 2, 3))
    dgamma = prod.sum(axis=(0, 2, 3))
    dx = np.empty(x.shape, dtype=F32)
    o2 = dx.reshape(N * C, H, W)
    gi2, db2, dg2 = rows(gamma * invstd), rows(dbeta), rows(dgamma)

    def grad(lo, hi):
        o2[lo:hi] = (gi2[lo:hi] * (d2[lo:hi] - db2[lo:hi] / m - h2[lo:hi] * dg2[lo:hi] / m)).astype(F32)

    _par(N * C, grad, nbytes=xd.nbytes)
    return dx, dgamma.astype(F32), dbeta.astype(F32)


def relu(x):
    """tensor.nnet.relu == 0.5*(x+|x|) (denet/layer/activation.py:31-34; batch_norm_relu.py:34-39)"""
    if not x.flags.c_contiguous:
        return ((x + np.abs(x)) / F32(2)).astype(x.dtype)
    y = np.empty(x.shape, dtype=x.dtype)
    xf, yf = x.reshape(-1), y.reshape(-1)

    def act(lo, hi):
        yf[lo:hi] = ((xf[lo:hi] + np.abs(xf[lo:hi])) / F32(2)).astype(x.dtype)

    _par(xf.size, act, nbytes=x.nbytes)
    return y


def relu_grad(y, dy):
    if not (dy.flags.c_contiguous and y.flags.c_contiguous and y.shape == dy.shape):
        return np.where(y > 0, dy, 0).astype(dy.dtype)
    out = np.empty(dy.shape, dtype=dy.dtype)
    yf, df, of = y.reshape(-1), dy.reshape(-1), out.reshape(-1)

    def mask(lo, hi):
        of[lo:hi] = np.where(yf[lo:hi] > 0, df[lo:hi], 0).astype(dy.dtype)

    _par(df.size, mask, nbytes=dy.nbytes)
    return out


# ---------------------------------------------------------------------------------------------------------
# pooling: denet/layer/pool.py:28-40 (cuDNN max pooling pads with -inf; average_inc_pad divides by k*k)
# ---------------------------------------------------------------------------------------------------------
def pool_max(x, k, s, p):
    N, C, H, W = x.shape
    OH = (H + 2 * p - k) // s + 1
    OW = (W + 2 * p - k) // s + 1
    xp = np.full((N, C, H + 2 * p, W + 2 * p), -np.inf, dtype=x.dtype)
    xp[:, :, p:p + H, p:p + W] = x
    win = np.empty((k * k, N, C, OH, OW), dtype=x.dtype)
    for a in range(k):
        for b in range(k):
            win[a * k + b] = xp[:, :, a:a + s * OH:s, b:b + s * OW:s]
    arg = win.argmax(axis=0)  # first maximum in (ky, kx) scan order
    return win.max(axis=0), arg


def pool_max_grad(dy, arg, x_shape, k, s, p):
    N, C, H, W = x_shape
    OH, OW = dy.shape[2], dy.shape[3]
    dxp = np.zeros((N, C, H + 2 * p, W + 2 * p), dtype=np.float64)
    for a in range(k):
        for b in range(k):
            sel = (arg == a * k + b)
            dxp[:, :, a:a + s * OH:s, b:b + s * OW:s] += np.where(sel, dy, 0)
    return dxp[:, :, p:p + H, p:p + W].astype(F32)


def pool_avg(x, k, s, p):
    N, C, H, W = x.shape
    OH = (H + 2 * p - k) // s + 1
    OW = (W + 2 * p - k) // s + 1
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (p, p), (p, p)))
    acc = np.zeros((N, C, OH, OW))
    for a in range(k):
        for b in range(k):
            acc += xp[:, :, a:a + s * OH:s, b:b + s * OW:s]
    return (acc / (k * k)).astype(F32)


def pool_avg_grad(dy, x_shape, k, s, p):
    N, C, H, W = x_shape
    OH, OW = dy.shape[2], dy.shape[3]
    dxp = np.zeros((N, C, H + 2 * p, W + 2 * p))
    for a in range(k):
        for b in range(k):
            dxp[:, :, a:a + s * OH:s, b:b + s * OW:s] += dy / (k * k)
    return dxp[:, :, p:p + H, p:p + W].astype(F32)


def pool_inv(x, size):
    """denet/layer/pool_inv.py:26: repeat(repeat(x, size[1], axis=2), size[0], axis=3)"""
    return np.repeat(np.repeat(x, size[1], axis=2), size[0], axis=3)


def pool_inv_grad(dy, size):
    """denet/layer/pool_inv_op.py:144-169: sum of each size[1] x size[0] block"""
    N, C, OH, OW = dy.shape
    return dy.reshape(N, C, OH // size[1], size[1], OW // size[0], size[0]).astype(np.float64).sum(axis=(3, 5)).astype(F32)


# ---------------------------------------------------------------------------------------------------------
# log-softmax and the DeNet corner head: denet/common/theano_util.py:27-29, denet/layer/denet_corner.py:50-53,126-134
# ---------------------------------------------------------------------------------------------------------
def log_softmax(x, axis):
    xdev = x - x.max(axis=axis, keepdims=True)
    return xdev - np.log(np.sum(np.exp(xdev), axis=axis, keepdims=True))


def corner_pr(corner_logits):
    """corner_logits (B,Cn,H,W) -> (B,2,Cn,H,W): lh = concat([x, -x]) on a new axis 1, log_softmax over it"""
    lh = np.stack([corner_logits, -corner_logits], axis=1).astype(F32)
    return log_softmax(lh, axis=1).astype(F32)


def corner_cost(target, pr, cost_factor, want_grad=True):
    """cost = cost_factor * -sum(target*pr, axes 1..4).mean() / ln 2 ; gradient w.r.t. the corner logits x"""
    B = pr.shape[0]
    t, l = target.astype(np.float64), pr.astype(np.float64)
    cost = cost_factor * (-(t * l).sum(axis=(1, 2, 3, 4)).mean() / math.log(2))
    grad = None
    if want_grad:
        p0, p1 = np.exp(l[:, 0]), np.exp(l[:, 1])
        T = t[:, 0] + t[:, 1]
        grad = (cost_factor / (B * math.log(2))) * (T * (p0 - p1) - (t[:, 0] - t[:, 1]))
        grad = grad.astype(F32)
    return float(cost), grad


def corner_target(metas, corner_shape, dropout=0.0):
    """denet/layer/denet_corner.py:81-123 (Python round = banker's rounding; x1 = max(x0, round(b2*W)-1))"""
    B, _, Cn, H, W = corner_shape
    t = np.zeros(corner_shape, dtype=F32)
    for b, meta in enumerate(metas):
        for bbox in meta["bbox"]:
            x0 = int(round(bbox[0] * W))
            y0 = int(round(bbox[1] * H))
            x1 = max(x0, int(round(bbox[2] * W)) - 1)
            y1 = max(y0, int(round(bbox[3] * H)) - 1)
            vx0, vy0 = 0 <= x0 < W, 0 <= y0 < H
            vx1, vy1 = 0 <= x1 < W, 0 <= y1 < H
            if vx0 and vy0:
                t[b, 1, 0, y0, x0] = 1.0
            if vx1 and vy0:
                t[b, 1, 1, y0, x1] = 1.0
            if vx0 and vy1:
                t[b, 1, 2, y1, x0] = 1.0
            if vx1 and vy1:
                t[b, 1, 3, y1, x1] = 1.0
            if Cn == 5:
                cx = int(round((bbox[0] + bbox[2]) * 0.5 * W))
                cy = int(round((bbox[1] + bbox[3]) * 0.5 * H))
                if 0 <= cx < W and 0 <= cy < H:
                    t[b, 1, 4, cy, cx] = 1.0
    t[:, 0] = 1.0 - t[:, 1]
    t /= W * H * Cn
    return t


# ---------------------------------------------------------------------------------------------------------
# sparse RoI sampling: Theano fallback denet/layer/denet_sparse.py:70-96 (rule 0) and the CUDA op
# denet/layer/denet_sparse_op.py:42-85 (rule 1); gradient denet_sparse_op.py:171-212 (scatter add).
#   rule 0: p = p0 + (i*extent)/(gs-1) in float32, clamp(p*size, 0, size-1), round half to even
#           (theano.tensor.round default since Theano 0.9, config `warn.round`)
#   rule 1: p = p0 + i*extent*k, k = 1.0f/(gs-1), lroundf (half away from zero)
# ---------------------------------------------------------------------------------------------------------
def sparse_taps(bbox, gs, H, W, rule=0):
    """bbox (M,4) float32 -> ys, xs int arrays (M, gs). Index work: ALWAYS evaluated in float32 as the reference does (I32 below),
    also when the float64 arbiter (oracle/model.py: float64_arbiter) widens the activations"""
    I32 = np.float32
    bbox = bbox.astype(I32)
    x0, y0, x1, y1 = bbox[:, 0], bbox[:, 1], bbox[:, 2], bbox[:, 3]
    bw, bh = (x1 - x0).astype(I32), (y1 - y0).astype(I32)
    ar = np.arange(gs, dtype=I32)

    def pos(p0, ext, size):
        if rule == 0:
            p = p0[:, None] + ((ar[None, :] * ext[:, None]).astype(I32) / I32(gs - 1)).astype(I32)
        else:
            k = I32(I32(1.0) / I32(gs - 1))
            p = p0[:, None] + ((ar[None, :] * ext[:, None]).astype(I32) * k).astype(I32)
        f = (p.astype(I32) * I32(size)).astype(I32)
        f = np.maximum(I32(0), np.minimum(f, I32(size - 1)))
        if rule == 0:
            return np.rint(f).astype(np.int64)
        return np.floor(f + I32(0.5)).astype(np.int64)   # f >= 0: half away from zero

    return pos(y0, bh, H), pos(x0, bw, W), bh, bw


def sparse_sample(fmap, bbox, gs, rule=0):
    """fmap (B,F,H,W), bbox (B,sn,sn,4) -> (B, gs*gs*F+2, sn, sn); channel (yi*gs+xi)*F+f; then h, w"""
    B, Fc, H, W = fmap.shape
    sn = bbox.shape[1]
    bb = bbox.reshape(B * sn * sn, 4)
    ys, xs, bh, bw = sparse_taps(bb, gs, H, W, rule)
    out = np.zeros((B, sn * sn, gs * gs * Fc + 2), dtype=F32)
    for b in range(B):
        sl = slice(b * sn * sn, (b + 1) * sn * sn)
        yy = ys[sl][:, :, None].repeat(gs, 2)          # (S, gs, gs)
        xx = xs[sl][:, None, :].repeat(gs, 1)
        v = fmap[b][:, yy, xx]                         # (F, S, gs, gs)
        out[b, :, :gs * gs * Fc] = v.transpose(1, 2, 3, 0).reshape(sn * sn, gs * gs * Fc)
        out[b, :, gs * gs * Fc] = bh[sl]
        out[b, :, gs * gs * Fc + 1] = bw[sl]
    return out.reshape(B, sn, sn, -1).transpose(0, 3, 1, 2), (ys, xs)


def sparse_sample_grad(dy, taps, fmap_shape, gs):
    """dy (B, gs*gs*F+2, sn, sn) -> d_fmap (B,F,H,W), accumulated in float64 (the reference uses fp32 atomics)"""
    B, Fc, H, W = fmap_shape
    ys, xs = taps
    sn = dy.shape[2]
    d = dy.transpose(0, 2, 3, 1).reshape(B, sn * sn, -1)[:, :, :gs * gs * Fc].reshape(B, sn * sn, gs, gs, Fc)
    out = np.zeros((B, H, W, Fc), dtype=np.float64)
    for b in range(B):
        sl = slice(b * sn * sn, (b + 1) * sn * sn)
        yy = ys[sl][:, :, None].repeat(gs, 2).reshape(-1)
        xx = xs[sl][:, None, :].repeat(gs, 1).reshape(-1)
        np.add.at(out[b], (yy, xx), d[b].reshape(-1, Fc))
    return out.transpose(0, 3, 1, 2).astype(F32)


# ---------------------------------------------------------------------------------------------------------
# RoI list editing and targets: denet/layer/denet_sparse.py:164-206, denet/layer/denet_detect.py:147-235
# (loops kept exactly as the reference writes them)
# ---------------------------------------------------------------------------------------------------------
def edit_samples(sample_bboxs, metas, sample_count, random_sample, sample_gt=True):
    import random
    for b, meta in enumerate(metas):
        n = sample_count - math.floor(random_sample * sample_count)
        if len(sample_bboxs[b]) > n:
            sample_bboxs[b] = random.sample(sample_bboxs[b], n)
        while len(sample_bboxs[b]) < sample_count:
            x0 = random.uniform(0.0, 1.0)
            y0 = random.uniform(0.0, 1.0)
            x1 = random.uniform(x0, 1.0)
            y1 = random.uniform(y0, 1.0)
            sample_bboxs[b].append((0.0, (x0, y0, x1, y1)))
        if sample_gt:
            for index, bbox in enumerate(meta["bbox"]):
                sample_bboxs[b][-(index + 1)] = (1.0, bbox)
    return sample_bboxs


def bbox_array(sample_bboxs, batch_size, sample_num):
    """denet/layer/denet_sparse.cc:670-699"""
    out = np.zeros((batch_size, sample_num, sample_num, 4), dtype=F32)
    for b, samples in enumerate(sample_bboxs):
        for i, s in enumerate(samples):
            for n in range(4):
                out[b, i // sample_num, i % sample_num, n] = s[1][n]
    return out


def overlap_iou(obj_bboxs, sample_bboxs):
    """denet/common/theano_util.py:38-59 in float32"""
    x = np.array(obj_bboxs, dtype=F32)
    y = np.array(sample_bboxs, dtype=F32)
    x_area = (x[:, 2] - x[:, 0]) * (x[:, 3] - x[:, 1])
    y_area = (y[:, 2] - y[:, 0]) * (y[:, 3] - y[:, 1])
    dx = np.maximum(np.minimum(x[:, None, 2], y[None, :, 2]) - np.maximum(x[:, None, 0], y[None, :, 0]), F32(0))
    dy = np.maximum(np.minimum(x[:, None, 3], y[None, :, 3]) - np.maximum(x[:, None, 1], y[None, :, 1]), F32(0))
    ai = dx * dy
    au = (x_area[:, None] + y_area[None, :] - ai)
    return ai / au


def detect_target(metas, sample_bbox_list, batch_size, sample_num, class_num, thresholds, use_bbox_reg=True,
                  use_jointfit=False, use_indfit=False):
    """denet/layer/denet_detect.py:147-235, loop for loop; with use_indfit a fourth array, the independent fitness
    target (B, 6, sn, sn) of :188-192 / :219-226, is returned as well"""
    t0, t1 = thresholds
    fitness_num = 5 if use_jointfit else 6
    null_class = class_num * fitness_num if use_jointfit else class_num
    s0 = null_class + 1
    det_pr = np.zeros((batch_size, s0, sample_num, sample_num), dtype=F32)
    det_pr[:, null_class, ...] = 1.0
    bbox_valid = np.zeros((batch_size, sample_num, sample_num), dtype=F32)
    bbox_reg = np.zeros((batch_size, 8, sample_num, sample_num), dtype=F32)
    bbox_reg[:, 2, ...] = 1.0
    bbox_reg[:, 3, ...] = 1.0
    bbox_reg[:, 6, ...] = 1.0
    bbox_reg[:, 7, ...] = 1.0
    indfit_pr = np.zeros((batch_size, fitness_num, sample_num, sample_num), dtype=F32)
    indfit_pr[:, 0, ...] = 1.0
    for b, meta in enumerate(metas):
        samples = [bbox for _, bbox in sample_bbox_list[b]]
        if len(meta["bbox"]) > 0 and len(samples) > 0:
            overlap = overlap_iou(meta["bbox"], samples)
            bbox_indexs, sample_indexs = np.where(overlap > t0)
            for obj, index in zip(bbox_indexs.tolist(), sample_indexs.tolist()):
                sample_i = index % sample_num
                sample_j = index // sample_num
                sample_cls = meta["class"][obj]
                sample_f = (float(overlap[obj, index]) - t0) / (1.0 - t0)
                if use_jointfit:
                    f = max(0, min(int(fitness_num * sample_f), fitness_num - 1))
                    det_pr[b, sample_cls * fitness_num + f, sample_j, sample_i] = 1.0
                else:
                    det_pr[b, sample_cls, sample_j, sample_i] = 1.0
                det_pr[b, null_class, sample_j, sample_i] = 0.0
                if use_indfit:
                    f = 1 + int(math.floor((fitness_num - 1) * sample_f))
                    f = max(1, min(f, fitness_num - 1))
                    indfit_pr[b, 0, sample_j, sample_i] = 0.0
                    indfit_pr[b, f, sample_j, sample_i] = 1.0
            if use_bbox_reg:
                overlap_max = overlap.argmax(axis=0)
                for index in range(len(samples)):
                    obj = overlap_max[index]
                    if overlap[obj, index] <= t1:
                        continue
                    sample = samples[index]
                    target = meta["bbox"][obj]
                    si, sj = index % sample_num, index // sample_num
                    bbox_valid[b, sj, si] = 1.0
                    bbox_reg[b, 0, sj, si] = 0.5 * (target[0] + target[2])
                    bbox_reg[b, 1, sj, si] = 0.5 * (target[1] + target[3])
                    bbox_reg[b, 2, sj, si] = target[2] - target[0]
                    bbox_reg[b, 3, sj, si] = target[3] - target[1]
                    bbox_reg[b, 4, sj, si] = 0.5 * (sample[0] + sample[2])
                    bbox_reg[b, 5, sj, si] = 0.5 * (sample[1] + sample[3])
                    bbox_reg[b, 6, sj, si] = sample[2] - sample[0]
                    bbox_reg[b, 7, sj, si] = sample[3] - sample[1]
    det_pr /= det_pr.sum(axis=1)[:, None, ...]
    nfactor = sample_num * sample_num
    det_pr /= nfactor
    bbox_valid /= nfactor
    if use_indfit:
        indfit_pr /= indfit_pr.sum(axis=1)[:, None, ...]
        indfit_pr /= nfactor
        return det_pr, bbox_valid, bbox_reg, indfit_pr
    return det_pr, bbox_valid, bbox_reg


def smooth_l1(x):
    """denet/common/theano_util.py:32-34"""
    a = np.abs(x)
    return np.where(a < 1, 0.5 * x * x, a - 0.5)


def detect_cost(out, det_t, bbox_valid, bbox_reg_t, sample_bbox, class_outputs, cost_factor, bbox_factor,
                use_bounded_iou=False, want_grad=True):
    """denet/layer/denet_detect.py:238-313. out (B, s0[+4], sn, sn) = the conv output.
    returns (det_cost, bbox_cost, d_out); note bbox_factor enters twice (:295 and :310)."""
    B = out.shape[0]
    s0 = class_outputs
    o = out.astype(np.float64)
    lp = log_softmax(o[:, :s0], axis=1)
    det_errors = -(det_t * lp).sum(axis=1) / math.log(s0)
    det_cost = cost_factor * det_errors.sum() / B
    d_out = np.zeros_like(o)
    if want_grad:
        T = det_t.sum(axis=1, keepdims=True)
        d_out[:, :s0] = (cost_factor / B / math.log(s0)) * (T * np.exp(lp) - det_t)
    bbox_cost = 0.0
    if bbox_factor > 0.0:
        reg = o[:, s0:s0 + 4]
        bt = bbox_reg_t.astype(np.float64)
        tgt, smp = bt[:, 0:4], bt[:, 4:8]
        if not use_bounded_iou:
            t = np.stack([(tgt[:, 0] - smp[:, 0]) / smp[:, 2], (tgt[:, 1] - smp[:, 1]) / smp[:, 3],
                          np.log(tgt[:, 2] / smp[:, 2]), np.log(tgt[:, 3] / smp[:, 3])], axis=1)
            d = t - reg
            dd = -np.ones_like(d)
        else:
            sb = sample_bbox.astype(np.float64)           # (B,sn,sn,4)
            scx, scy = 0.5 * (sb[..., 0] + sb[..., 2]), 0.5 * (sb[..., 1] + sb[..., 3])
            sw, sh = sb[..., 2] - sb[..., 0], sb[..., 3] - sb[..., 1]
            pcx, pcy = reg[:, 0] * sw + scx, reg[:, 1] * sh + scy
            pw, ph = np.exp(reg[:, 2]) * sw, np.exp(reg[:, 3]) * sh
            px0, py0, px1, py1 = pcx - pw * 0.5, pcy - ph * 0.5, pcx + pw * 0.5, pcy + ph * 0.5
            predict_x, predict_y = 0.5 * (px0 + px1), 0.5 * (py0 + py1)
            predict_w, predict_h = px1 - px0, py1 - py0
            eps = 0.001
            dx, dyv = tgt[:, 0] - predict_x, tgt[:, 1] - predict_y
            cx = np.where(dx >= 0, 2 * dx / (tgt[:, 2] + dx + eps), -2 * dx / (tgt[:, 2] - dx + eps))
            cy = np.where(dyv >= 0, 2 * dyv / (tgt[:, 3] + dyv + eps), -2 * dyv / (tgt[:, 3] - dyv + eps))
            aw, bw_ = tgt[:, 2] / (predict_w + eps), predict_w / (tgt[:, 2] + eps)
            ah, bh_ = tgt[:, 3] / (predict_h + eps), predict_h / (tgt[:, 3] + eps)
            cw, ch = 1.0 - np.minimum(aw, bw_), 1.0 - np.minimum(ah, bh_)
            d = np.stack([cx, cy, cw, ch], axis=1)
            ddx = np.where(dx >= 0, 2 * (tgt[:, 2] + eps) / (tgt[:, 2] + dx + eps) ** 2,
                           -2 * (tgt[:, 2] + eps) / (tgt[:, 2] - dx + eps) ** 2) * (-sw)
            ddy = np.where(dyv >= 0, 2 * (tgt[:, 3] + eps) / (tgt[:, 3] + dyv + eps) ** 2,
                           -2 * (tgt[:, 3] + eps) / (tgt[:, 3] - dyv + eps) ** 2) * (-sh)
            ddw = np.where(aw <= bw_, tgt[:, 2] / (pw + eps) ** 2 * pw, -pw / (tgt[:, 2] + eps))
            ddh = np.where(ah <= bh_, tgt[:, 3] / (ph + eps) ** 2 * ph, -ph / (tgt[:, 3] + eps))
            dd = np.stack([ddx, ddy, ddw, ddh], axis=1)
        bbox_errors = bbox_factor * bbox_valid * smooth_l1(d).sum(axis=1)
        bbox_cost = bbox_factor * bbox_errors.sum() / B
        if want_grad:
            dsl = np.where(np.abs(d) < 1, d, np.sign(d))
            d_out[:, s0:s0 + 4] = (bbox_factor * bbox_factor / B) * bbox_valid[:, None] * dsl * dd
    return float(det_cost), float(bbox_cost), d_out.astype(F32)


def indfit_cost(out, indfit_t, offset, indfit_factor):
    """independent fitness cost (denet_detect.py:103-108 log-softmax of the logits behind the box regressors, :298-301,
    :311-312): -sum(target * logp) / ln(n), x factor / batch. returns (cost, d_out with only its slice filled)"""
    B = out.shape[0]
    n = indfit_t.shape[1]
    o = out.astype(np.float64)
    lp = log_softmax(o[:, offset:offset + n], axis=1)
    err = -(indfit_t * lp).sum(axis=1) / math.log(n)
    cost = indfit_factor * err.sum() / B
    d_out = np.zeros_like(o)
    T = indfit_t.sum(axis=1, keepdims=True)
    d_out[:, offset:offset + n] = (indfit_factor / B / math.log(n)) * (T * np.exp(lp) - indfit_t)
    return float(cost), d_out.astype(F32)


def regression_cost(logits, classes, want_grad=True):
    """denet/layer/regression.py:97-98: -mean(log_softmax(x)[b, cls]) for x (B, C, 1, 1)"""
    B, C = logits.shape[0], logits.shape[1]
    x = logits.reshape(B, C).astype(np.float64)
    lp = log_softmax(x, axis=1)
    cost = -lp[np.arange(B), classes].mean()
    grad = None
    if want_grad:
        grad = np.exp(lp)
        grad[np.arange(B), classes] -= 1.0
        grad = (grad / B).reshape(logits.shape).astype(F32)
    return float(cost), grad


# ---------------------------------------------------------------------------------------------------------
# solver: denet/model/model_cnn.py:282-294, 321-331
# ---------------------------------------------------------------------------------------------------------
def adam_update(p, m, v, g, lr, betas, iteration, decay, is_weight):
    """model_cnn.py:296-305 (+ L2 :323-324); bias corrections in double, the state in float32"""
    p, m, v, g = p.astype(F32), m.astype(F32), v.astype(F32), g.astype(F32)
    if is_weight:
        g = g + F32(decay) * p
    b1, b2 = F32(betas[0]), F32(betas[1])
    m2 = b1 * m + (F32(1.0) - b1) * g
    v2 = b2 * v + (F32(1.0) - b2) * (g * g)
    c1 = F32(1.0 / (1.0 - float(b1) ** (iteration + 1)))
    c2 = F32(1.0 / (1.0 - float(b2) ** (iteration + 1)))
    p2 = p - F32(lr) * (m2 * c1) / (np.sqrt(v2 * c2) + F32(1e-8))
    return p2.astype(F32), m2.astype(F32), v2.astype(F32)


def solver_update(p, m, g, lr, momentum, iteration, decay, is_weight, mode="nesterov"):
    p, m, g = p.astype(F32), m.astype(F32), g.astype(F32)
    if is_weight:
        g = g + F32(decay) * p
    rho = F32(momentum if iteration > 0 else 0.0)
    if mode in ("torch", "nesterov"):
        m2 = rho * m + g
        p2 = p - F32(lr) * (g + F32(momentum) * m2)
    else:
        m2 = rho * m + (F32(1.0) - rho) * g
        p2 = p - F32(lr) * m2
    return p2.astype(F32), m2.astype(F32)


# ---------------------------------------------------------------------------------------------------------
# inference decode: denet/layer/denet_detect.py:76-100 (det_pr, bbox_predict), :330-349 (joint fitness)
# ---------------------------------------------------------------------------------------------------------
def detect_outputs(out, sample_bbox, class_num, jointfit, t0, use_bbox_reg=True, nfit=0):
    """out (B, s0[+4], sn, sn) -> det_pr (B,C+1,sn,sn), fitness (B,C+1,sn,sn), bbox (B,sn,sn,4)"""
    B, _, sn, _ = out.shape
    fit = 5
    s0 = class_num * fit + 1 if jointfit else class_num + 1
    o = out.astype(F32)
    lp = log_softmax(o[:, :s0], axis=1).astype(F32)
    if jointfit:
        det_fit = lp[:, :class_num * fit].reshape(B, class_num, fit, sn, sn)
        null = lp[:, class_num * fit]
        m = det_fit.max(axis=2)
        det_pr = m + np.log(np.sum(np.exp(det_fit - m[:, :, None]), axis=2))
        det_pr = np.concatenate([det_pr, null[:, None]], axis=1).astype(F32)
        val = np.array([t0 + i * (1.0 - t0) / fit for i in range(fit)], dtype=F32)
        fitness = np.log(np.sum(np.exp(det_fit) * val[None, None, :, None, None], axis=2))
        fitness = np.concatenate([fitness, null[:, None]], axis=1).astype(F32)
    else:
        det_pr = lp
        fitness = lp.copy()
    if use_bbox_reg:
        reg = o[:, s0:s0 + 4]
        sb = sample_bbox.astype(F32)
        scx, scy = F32(0.5) * (sb[..., 0] + sb[..., 2]), F32(0.5) * (sb[..., 1] + sb[..., 3])
        sw, sh = sb[..., 2] - sb[..., 0], sb[..., 3] - sb[..., 1]
        pcx, pcy = reg[:, 0] * sw + scx, reg[:, 1] * sh + scy
        pw, ph = np.exp(reg[:, 2]) * sw, np.exp(reg[:, 3]) * sh
        bbox = np.stack([pcx - pw * F32(0.5), pcy - ph * F32(0.5), pcx + pw * F32(0.5), pcy + ph * F32(0.5)], axis=-1)
    else:
        bbox = sample_bbox
    if nfit > 0:
        # denet_detect.py:396-401: fitness += log(sum_f indfit_pr[f] * val[f]) (expectation in double, cast to float32)
        off = s0 + (4 if use_bbox_reg else 0)
        ip = np.exp(log_softmax(o[:, off:off + nfit], axis=1).astype(F32))
        val = np.array([0.0] + [t0 + i * (1.0 - t0) / (nfit - 1) for i in range(nfit - 1)])
        fexp = np.sum(ip * val[None, :, None, None], axis=1).astype(F32)
        fitness = fitness + np.log(fexp)[:, None, :, :]
    return det_pr, fitness.astype(F32), bbox.astype(F32)
