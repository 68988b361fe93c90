"""`BN` layer — spatial batch normalisation. Mirrors denet/layer/batch_norm.py (BatchNormLayer :12-129):
cuDNN batch statistics (biased variance, eps inside the sqrt), running `mean` and running INVERSE standard
deviation `stdinv` with momentum (:75-76), gamma/beta reported as biases (no L2 decay, :106-107), and the
test-time path that feeds var = 1/stdinv^2 to cuDNN which adds eps a second time (:50-52)."""
import numpy

from . import AbstractLayer, Act, Param, get_train
from .. import ops


class BatchNormLayer(AbstractLayer):
    type_name = "batchnorm"
    fused_relu = False

    def __init__(self, layers, momentum=0.9, eps=1e-5, renorm_max_r=1.0, renorm_max_d=0.0, renorm_max_it=10,
                 json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.enabled = json_param.get("enabled", True)
        self.momentum = json_param.get("momentum", momentum)
        self.renorm_max_r = json_param.get("renormMaxR", renorm_max_r)
        self.renorm_max_d = json_param.get("renormMaxD", renorm_max_d)
        self.renorm_max_it = json_param.get("renormMaxIt", renorm_max_it)
        self.eps = json_param.get("eps", eps)
        self.output_shape = self.input_shape
        if self.enabled:
            assert self.input.cp == self.input_shape[1], "batch norm needs an unpadded channel count (multiple of 32)"
            c = self.input_shape[1]
            self.omega = Param(numpy.ones((c,)), "bn omega")
            self.beta = Param(numpy.zeros((c,)), "bn beta")
            self.mean = Param(numpy.zeros((c,)), "bn mean")
            self.stdinv = Param(numpy.ones((c,)), "bn std inv")
            self.output = Act(self.output_shape, self.input.cp, "bn%i" % self.layer_index)
            self.input.want_stats = True      # a convolution writing this tensor also emits its per-channel sums
            self.input.stats_bn = self        # ... and may finish this layer's statistics in its own launch (ops.BnFinal)
        else:
            self.output = self.input
        self._save = None

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "BN":
            return False
        layers.append(BatchNormLayer(layers, params.get(0, 0.9), params.get(1, 1e-5), params.get(2, 1),
                                     params.get(3, 0), params.get(4, 0)))
        return True

    def params(self):
        return [self.omega, self.beta, self.mean, self.stdinv] if self.enabled else []

    def updates(self, cost=None):
        return [self.mean, self.stdinv] if self.enabled else []

    def biases(self):
        return [self.omega, self.beta] if self.enabled else []

    def all_params(self):
        return [self.omega, self.beta] if self.enabled else []

    def export_json(self):
        json = super().export_json()
        json.update({"momentum": self.momentum,
                     "eps": self.eps,
                     "mean": self.mean.get_value() if self.enabled else None,
                     "std": self.stdinv.get_value() if self.enabled else None,
                     "gamma": self.omega.get_value() if self.enabled else None,
                     "bias": self.beta.get_value() if self.enabled else None,
                     "renormMaxR": self.renorm_max_r,
                     "renormMaxD": self.renorm_max_d,
                     "renormMaxIt": self.renorm_max_it,
                     "enabled": self.enabled})
        return json

    def import_json(self, json_param):
        if self.enabled:
            self.omega.set_value(numpy.asarray(json_param["gamma"], dtype=numpy.float32))
            self.beta.set_value(numpy.asarray(json_param["bias"], dtype=numpy.float32))
            self.mean.set_value(numpy.asarray(json_param["mean"], dtype=numpy.float32))
            self.stdinv.set_value(numpy.asarray(json_param["std"], dtype=numpy.float32))

    # ---- execution ----
    def stats_final(self, act):
        """for the convolution pass that writes `act` (this layer's input) in a training step: the description of this layer's
        statistics, so that the pass can finish them in its own launch (ops.BnFinal; the running statistics are updated there)"""
        if not (self.enabled and ops.FINAL_FOLD and act is self.input):
            return None
        if getattr(self, "pool_behind", None) is not None and ops.BN_POOL_FUSE:
            # the fused BN + ReLU + max-pool pass reduces the partial rows itself and updates the running statistics there
            # (ops.bn_relu_pool_fwd_train reads only the rows): a producer that also finished them would advance run_mean /
            # run_stdinv twice per step. The producer keeps writing rows only.
            return None
        n, c = act.shape[0], act.cp
        m = 1
        for d in act.shape[2:]:
            m *= d
        return ops.BnFinal(1, n * m, c, self.__dict__.setdefault("_bnf_holder_fwd", {}), momentum=self.momentum, eps=self.eps,
                           run_mean=self.mean.dev, run_stdinv=self.stdinv.dev)

    def forward(self, ctx, res=None, relu=None, out_act=None):
        """res / relu / out_act let ResnetLayer fuse the residual add + ReLU into the normalisation pass"""
        if not self.enabled:
            return
        if relu is None and out_act is None and getattr(self, "act_fused", False):
            relu, out_act = True, self.act_behind.output          # `BN A`: see ActivationLayer
        relu = self.fused_relu if relu is None else relu
        out_act = self.output if out_act is None else out_act
        x = self.input.data
        if get_train():
            # per-channel sums already written by the convolution that produced x (ConvLayer.forward), valid for this tensor only
            pre = getattr(self.input, "stats", None)
            self.input.stats = None
            pool = getattr(self, "pool_behind", None)
            own_out = self.act_behind.output if getattr(self, "act_fused", False) else self.output
            if pool is not None and ctx is not None and relu and res is None and out_act is own_out and ops.BN_POOL_FUSE:
                # a max pool is the only reader of this layer's output (ModelCNN.build_train_func links it): one pass
                # writes the pooled tensor, relu(bn(x)) itself is never materialised (its gradient neither)
                k, s, p = pool.size[0], pool.stride[0], pool.pad[0]
                assert pre is None or len(pre) == 2, "statistics finished by the producer in front of a pool-fused batch norm"
                yp, arg, sm, si, xh = ops.bn_relu_pool_fwd_train(x, self.omega.dev, self.beta.dev, self.mean.dev, self.stdinv.dev,
                                                                 k, s, p, self.momentum, self.eps, pre=pre, xhat=True)
                pool.output.data, pool._arg, pool._fused_in = yp, arg, ctx      # valid for this pass (ctx) only
                # the backward reductions of this layer are sums over the pooled tensors (ops.bn_relu_pool_bwd_pooled); the
                # data-gradient pass that writes the pool output's gradient can leave them behind (sums_request)
                pool._xhat = xh
                pool.output.bn_producer = self
                out_act.data = None
                self._save = (sm, si, relu, out_act, False)
                self._pooled = True
                return
            self._pooled = False
            if pre is not None and ops.LINK_BN and ctx is not None:
                # the statistics are reduced now; the pointwise pass is left to the reader of the output (a Winograd convolution
                # runs it inside its input transform, anybody else triggers it by taking out_act.data - ops.BnLink)
                link, sm, si = ops.bn_fwd_train_link(x, self.omega.dev, self.beta.dev, self.mean.dev, self.stdinv.dev, pre,
                                                     self.momentum, self.eps, relu=relu, res=res)
                self._save = (sm, si, relu, out_act, res is not None)
                out_act.set_pending_data(link)
                out_act.bn_producer = self
                return
            y, sm, si = ops.bn_fwd_train(x, self.omega.dev, self.beta.dev, self.mean.dev, self.stdinv.dev,
                                         self.momentum, self.eps, relu=relu, res=res, pre=pre)
            self._save = (sm, si, relu, out_act, res is not None)
            out_act.data = y
            out_act.bn_producer = self
            return
        else:
            y = ops.bn_fwd_test(x, self.omega.dev, self.beta.dev, self.mean.dev, self.stdinv.dev, self.eps, relu=relu,
                                res=res, cache=self.__dict__.setdefault("_infer_cache", {}))
        out_act.data = y

    def sums_request(self, act):
        """for the data-gradient pass that writes the gradient of `act` (this layer's output of the current training step): the
        description of this batch norm, so that the pass leaves the two backward reductions behind (ops.BnSums)"""
        if not (self.enabled and self._save is not None):
            return None
        if getattr(self, "_pooled", False):
            # the fused BN + ReLU + max pool: `act` is the POOL's output; over the pooled tensors the layer looks like a batch norm
            # with input xhat (already normalised: mean 0, invstd 1) and forward output y_pool
            pool = self.pool_behind
            if act is not pool.output or getattr(pool, "_xhat", None) is None:
                return None
            C = act.data.shape[-1]
            return ops.BnSums(pool._xhat, act.data, self.omega.dev, self.beta.dev, ops.const_vec(0.0, C), ops.const_vec(1.0, C), True)
        sm, si, relu, out_act, has_res = self._save
        if out_act is not act:
            return None
        # the data-gradient pass may also FINISH the two reductions (dgamma, dbeta, their means) in its own launch (ops.BnFinal)
        x = self.input.data
        fin = None
        if ops.FINAL_FOLD and self.omega.grad is not None and self.beta.grad is not None:
            fin = ops.BnFinal(2, x.numel() // x.shape[-1], x.shape[-1], self.__dict__.setdefault("_bnf_holder", {}),
                              dgamma=self.omega.grad, dbeta=self.beta.grad)
        return ops.BnSums(x, out_act.data if (relu and has_res) else None, self.omega.dev, self.beta.dev, sm, si, relu, final=fin)

    def backward(self, ctx, want_dres=False):
        if not self.enabled:
            return None
        sm, si, relu, out_act, has_res = self._save
        if getattr(self, "_pooled", False):
            pool = self.pool_behind
            k, s, p = pool.size[0], pool.stride[0], pool.pad[0]
            dyp = pool.output.grad
            sums = pool.output.grad_sums       # left by the data-gradient pass that wrote dyp last (ConvLayer.backward)
            dx = ops.bn_relu_pool_bwd_pooled(self.input.data, pool._xhat, pool.output.data, dyp, pool._arg, self.omega.dev,
                                             self.beta.dev, sm, si, k, s, p, self.omega.grad, self.beta.grad,
                                             pre=sums.partial if sums is not None else None)
            pool._xhat = None
            self.input.add_grad(dx)
            return None
        # without a residual input the relu mask is recomputed from x in the kernel (no read of y)
        y = out_act.data if (relu and has_res) else None
        dy = out_act.grad
        sums = out_act.grad_sums           # left by the data-gradient pass that wrote dy last (ConvLayer.backward)
        pre = sums.partial if sums is not None else None
        if pre is not None and not (ops.LINK_BN and self.input._grad is None and self.input._pending_grad is None):
            link, dres = ops.bn_bwd_link(self.input.data, y, dy, self.omega.dev, sm, si, relu=relu, want_dres=want_dres,
                                         dgamma=self.omega.grad, dbeta=self.beta.grad, beta=self.beta.dev, pre=pre)
            self.input.add_grad(link.materialise())
            return dres
        if ops.LINK_BN and self.input._grad is None and self.input._pending_grad is None:
            # the two sums now; the gradient of the input is formed by whoever reads it - the convolution in front inside the
            # transform of its data- and filter-gradient passes (ConvLayer.backward), anybody else through Act.grad
            link, dres = ops.bn_bwd_link(self.input.data, y, dy, self.omega.dev, sm, si, relu=relu,
                                         want_dres=want_dres, dgamma=self.omega.grad, dbeta=self.beta.grad, beta=self.beta.dev,
                                         pre=pre)
            self.input.set_pending_grad(link)
            return dres
        dx, dres, _, _ = ops.bn_bwd(self.input.data, y, dy, self.omega.dev, sm, si, relu=relu,
                                    want_dres=want_dres, dgamma=self.omega.grad, dbeta=self.beta.grad,
                                    beta=self.beta.dev)
        self.input.add_grad(dx)
        return dres


def test():
    """The reference's known-answer test (denet/layer/batch_norm.py:131-154), runnable on a GPU box."""
    from . import InitialLayer, set_train
    import torch
    numpy.random.seed(1002)
    eps = 1e-4
    input_shape = (64, 128, 32, 32)
    bn = BatchNormLayer([InitialLayer(Act(input_shape), input_shape)])
    for p in bn.params():
        p.dev = torch.from_numpy(p.to_dev_layout()).cuda()
    x = numpy.random.uniform(0.0, 1.0, input_shape).astype(numpy.float32)
    bn.input.data = ops.nchw_to_nhwc(torch.from_numpy(x).cuda(), 128)
    set_train(True)
    bn.forward(None)
    y = bn.output.data
    x_mean = bn.mean.get_value()
    x_std = bn.stdinv.get_value()
    if abs(float(y.mean())) > eps or abs(float(y.std()) - 1.0) > eps or abs(x_mean.mean() - x.mean() * 0.1) > eps \
            or abs(x_std.mean() - 1.24641) > eps:
        raise Exception("Batchnorm failed test! ", float(y.mean()), float(y.std()), x_mean.mean(), x_std.mean())
