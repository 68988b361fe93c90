"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL all-reduce over xGMI.

The reference's only parallelism is host-side PARAMETER AVERAGING: every worker copies all update targets
(parameters, momentum, BN running statistics) to host shared memory, the parent averages them with numpy and the
workers copy them back (denet/model/train_multi.py:96-145, denet/multi/shared.py:105-119,155-165,
denet/multi/worker.py:89-93). For the linear solvers (sgd, torch/nesterov) with identical initial state and
batch_size_factor=1 that equals: average the GRADIENTS, update once, and average the BN running statistics —
which is what this module does with `torch.distributed` (backend "nccl" is RCCL on ROCm):

  * gradients live in one flat buffer [weights in layer order | biases]; the weight region is cut into buckets
    of >= bucket_bytes; the all-reduce of a bucket is issued (async, on RCCL's stream) as soon as the backward
    sweep has passed the first layer of that bucket, so it overlaps the remaining backward kernels;
  * the 1/N of the mean is folded into the solver kernel (grad_scale), no extra pass over the gradients;
  * the bias / BN-affine gradients (complete only when the sweep is) and the BN running mean / stdinv (a few thousand
    floats, averaged like shared.py does) ride in the LAST bucket's collective - the one issued when the sweep has passed
    the first layer - through a packed copy: a step has exactly len(buckets) collectives and none is issued in finish_step.
"""
import os


class DataParallel:
    def __init__(self, backend=None, bucket_bytes=32 << 20, scale_fn=None, init=True):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.bucket_bytes = int(bucket_bytes)
        if init and not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend == "nccl":
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group(backend=backend)
            if backend == "nccl":
                # RCCL and the process group have taken their hardware queues: place the side streams now
                import torch.distributed as _d
                t = torch.zeros(1, device="cuda")
                _d.all_reduce(t)
                from .. import ops
                ops.init_streams(force=True)
        self.backend = dist.get_backend()
        self.world_size = dist.get_world_size()
        self.rank = dist.get_rank()
        self.scale_fn = scale_fn
        self.force_collectives = False     # run the collectives even at world_size 1 (single-GPU smoke of the path)
        self._buckets = None
        self._pending = []
        self._timing = None                # [(event before the wait, event after)] per step while start_timing() is on
        self._step_bytes = 0
        self._step_colls = 0

    # ---- instrumentation (bench.py) ----------------------------------------------------------------------
    def start_timing(self):
        """from now on finish_step brackets its wait for the collectives with two HIP events on the compute stream, and every
        bucket leaves three events: issued (on the filter-gradient stream, where its collective is queued), the end of the backward
        sweep, and the point at which the compute stream had waited for it"""
        self._timing = []
        self._bucket_timing = []           # per step: [(bytes, issued event, sweep-end event, waited-for event)] per bucket

    def bucket_overlap_table(self):
        """per bucket, averaged over the timed steps: bytes, how long BEFORE the end of the backward sweep its collective was issued
        (the window the exchange has to hide in), and how long AFTER the end of the sweep the compute stream still waited for it
        (what the backward pass did not hide). The first real multi-GPU run is read against this table."""
        if not getattr(self, "_bucket_timing", None):
            return []
        import torch
        torch.cuda.synchronize()
        n = len(self._bucket_timing)
        rows = None
        for step in self._bucket_timing:
            if rows is None:
                rows = [{"bytes": b, "issued_before_sweep_end_ms": 0.0, "waited_after_sweep_end_ms": 0.0} for b, _, _, _ in step]
            for r, (_, e_issue, e_end, e_done) in zip(rows, step):
                r["issued_before_sweep_end_ms"] += e_issue.elapsed_time(e_end) / n
                r["waited_after_sweep_end_ms"] += max(0.0, e_end.elapsed_time(e_done)) / n
        for r in rows or []:
            r["issued_before_sweep_end_ms"] = round(r["issued_before_sweep_end_ms"], 3)
            r["waited_after_sweep_end_ms"] = round(r["waited_after_sweep_end_ms"], 3)
        return rows or []

    def exposed_ms_per_step(self):
        """mean time the compute stream stood in finish_step's wait: collective time the backward pass did not hide"""
        if not self._timing:
            return 0.0
        import torch
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._timing) / len(self._timing)

    def bytes_per_step(self):
        return int(self._step_bytes)

    def collectives_per_step(self):
        return int(self._step_colls)

    def gather_floats(self, value):
        """[value of rank 0, value of rank 1, ...] on every rank"""
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        mine = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        if self.world_size == 1:
            return [float(value)]
        got = [torch.zeros_like(mine) for _ in range(self.world_size)]
        self.dist.all_gather(got, mine)
        return [float(t.item()) for t in got]

    def _all_reduce(self, t):
        self._step_bytes += t.numel() * t.element_size()
        self._step_colls += 1
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=True)

    # ---- setup -------------------------------------------------------------------------------------------
    def broadcast_state(self, model):
        """identical initial state on every rank (reference: model_write of the shared state to every worker,
        train_multi.py:102, shared.py:90-92)"""
        for t in self._state(model):
            self.dist.broadcast(t, src=0)
        import torch
        if torch.cuda.is_available():
            from .. import ops
            ops.bump_weights_version()

    @staticmethod
    def _state(model):
        """every update target the reference averages (shared.py:105-119 walks model.updates: parameters, momentum, the
        adam second moments once they exist, BN running statistics)"""
        v = getattr(model, "V", None)
        return [model.P, model.M, model.S] + ([v] if v is not None else [])

    def make_buckets(self, layer_weight_range):
        """layer_weight_range: [(layer, lo, hi)] in layer order. Returns [(lo, hi, trigger_layer)]: contiguous
        slices of the weight region, built from the LAST layer backwards; a bucket is complete once the backward
        sweep has finished `trigger_layer` (its lowest-index layer). DeNet-34 skip at the default 32 MB: 40 / 34 / 33 / 24 MB
        (head, up path + stage 4, stage 4-3, stage 3) and a 6 MB remainder (stages 1-2, stem) for the exposed tail."""
        buckets = []
        cur_hi = None
        cur_lo = None
        trigger = None
        min_elems = max(1, self.bucket_bytes // 4)
        tail_elems = max(1, min_elems // 4)
        for layer, lo, hi in reversed(layer_weight_range):
            if cur_hi is None:
                cur_hi = hi
            cur_lo = lo
            trigger = layer
            # close the bucket when it is full - or when what is left below it (the first layers of the network, whose
            # gradients arrive at the very end of the backward sweep) is small: the last collective, which nothing
            # overlaps, then carries only that remainder instead of a whole bucket
            if cur_hi - cur_lo >= min_elems or (0 < cur_lo <= tail_elems and cur_hi - cur_lo >= tail_elems):
                buckets.append((cur_lo, cur_hi, trigger))
                cur_hi = None
        if cur_hi is not None:
            buckets.append((cur_lo, cur_hi, trigger))
        return buckets

    # ---- per step ----------------------------------------------------------------------------------------
    def begin_step(self, model):
        if self._buckets is None:
            self._buckets = self.make_buckets(model.layer_weight_range)
            self._trigger = {}
            for lo, hi, layer in self._buckets:
                self._trigger[id(layer)] = (lo, hi)
        self._pending = []
        self._issue_events = []
        self._step_bytes = 0
        self._step_colls = 0

    def layer_done(self, model, layer):
        r = self._trigger.get(id(layer))
        if r is not None and (self.world_size > 1 or self.force_collectives):
            lo, hi = r
            # the bucket holds convolution weight gradients only: they are produced on the wgrad stream, so the
            # collective is ordered behind THAT stream and the data-gradient chain on the compute stream never waits
            with self._wgrad_ctx():
                if self._timing is not None:
                    import torch
                    e_issue = torch.cuda.Event(enable_timing=True)
                    e_issue.record()               # on the stream the collective is ordered behind
                    self._issue_events.append(e_issue)
                if lo == self._buckets[-1][0]:
                    # the sweep has passed the first layer: every gradient of the step is queued. The last collective carries
                    # [this bucket | bias and BN-affine gradients | BN running statistics] as one packed tensor
                    self._tail = self._pack_tail(model, lo, hi)
                    self._pending.append(self._all_reduce(self._tail[0]))
                else:
                    self._pending.append(self._all_reduce(model.G[lo:hi]))

    def _tail_parts(self, model, lo, hi):
        parts = [model.G[lo:hi]]
        if model.n_trainable > model.n_weights:
            parts.append(model.G[model.n_weights:model.n_trainable])
        parts.append(model.S)
        return parts

    def _pack_tail(self, model, lo, hi):
        import torch
        parts = self._tail_parts(model, lo, hi)
        n = sum(p.numel() for p in parts)
        buf = getattr(self, "_tail_buf", None)
        if buf is None or buf.numel() != n or buf.device != parts[0].device:
            buf = self._tail_buf = torch.empty(n, dtype=parts[0].dtype, device=parts[0].device)
        o = 0
        for p in parts:
            buf[o:o + p.numel()].copy_(p)
            o += p.numel()
        return buf, lo, hi

    @staticmethod
    def _wgrad_ctx():
        """ops.wgrad_stream() on a GPU (the stream the weight gradients are produced on), a no-op context on CPU"""
        import contextlib
        import torch
        if torch.cuda.is_available():
            from .. import ops
            return ops.wgrad_stream()
        return contextlib.nullcontext()

    def finish_step(self, model):
        if self.world_size > 1 or self.force_collectives:
            ev = None
            if self._timing is not None:
                import torch
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            done = []
            for w in self._pending:
                w.wait()
                if ev is not None:
                    import torch
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()                     # the compute stream has waited for this bucket's collective
                    done.append(e)
            if ev is not None and len(done) == len(self._issue_events) == len(self._buckets or []):
                self._bucket_timing.append([(4 * (hi - lo), ei, ev[0], ed)
                                            for (lo, hi, _), ei, ed in zip(self._buckets, self._issue_events, done)])
            tail = self.__dict__.pop("_tail", None)
            if tail is None:          # a model without bucketed weights: the packed collective is all there is
                tail = self._pack_tail(model, 0, 0)
                self._all_reduce(tail[0]).wait()
            buf, lo, hi = tail
            o = 0
            for p in self._tail_parts(model, lo, hi):
                p.copy_(buf[o:o + p.numel()])
                o += p.numel()
            self._scale(model.S, 1.0 / self.world_size)
            if ev is not None:
                ev[1].record()
                self._timing.append(ev)
        self._pending = []

    def _scale(self, t, s):
        if self.scale_fn is not None:
            self.scale_fn(t, s)
        else:
            from .. import ops
            ops.check(ops._L().denet_scale(t.data_ptr(), t.numel(), float(s), ops.stream_ptr()), "scale")

    def average_state(self, model):
        """the reference's own scheme (train_multi.py:96-145, shared.py:105-119): every rank has run its local training
        steps; parameters, momentum and BN running statistics are replaced by their mean over the ranks"""
        if self.world_size == 1 and not self.force_collectives:
            return
        d = self.dist
        state = self._state(model)
        work = [d.all_reduce(t, op=d.ReduceOp.SUM, async_op=True) for t in state]
        for w in work:
            w.wait()
        for t in state:
            self._scale(t, 1.0 / self.world_size)
        import torch
        if torch.cuda.is_available():
            from .. import ops
            ops.bump_weights_version()

    def barrier(self):
        self.dist.barrier()

    def max_over_ranks(self, value):
        """max of a python float over all ranks (bench timing)"""
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        if self.world_size > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())
