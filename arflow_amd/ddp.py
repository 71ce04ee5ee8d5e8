"""One-process-per-GPU data parallelism for the image-pair batch (RCCL over xGMI).

The reference only has single-process ``torch.nn.DataParallel`` (trainer/base_trainer.py:75): per
step it broadcasts the parameters, scatters the batch, gathers the flows to GPU 0 and reduce-adds
the gradients.  Here every rank owns a full replica and its shard of the pairs; the only
communication is the gradient all-reduce:

  * all gradients live in ONE flat fp32 buffer (9-29 MB for these models);
    ``param.grad`` are views into it, so there is no flatten/unflatten copy;
  * the buffer is cut into a few contiguous buckets in reverse registration order (the order
    backward produces gradients); a bucket's all-reduce is launched asynchronously from the
    post-accumulate-grad hook of its last parameter, overlapping the rest of backward;
  * on the 8-GPU xGMI mesh RCCL serves this size class with a direct reduce-scatter/all-gather
    (7 links in parallel), so few large buckets beat many small ones: per-link-bound, ~0.1 ms.

Works with ``backend='nccl'`` (= RCCL on ROCm) and with ``gloo`` on CPU (tests, world_size 2).
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, module, n_buckets=4, process_group=None, force_collectives=False):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # force_collectives: run the bucket all-reduces even at world size 1 (exercises the RCCL path on a
        # one-GPU box; a 1-rank all-reduce is the identity)
        self._active = self.world > 1 or (force_collectives and dist.is_available() and dist.is_initialized())
        params = [p for p in module.parameters() if p.requires_grad]
        assert params, 'no trainable parameters'
        dev, dt = params[0].device, params[0].dtype
        total = sum(p.numel() for p in params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        # reverse registration order ~ the order in which backward finishes gradients
        order = list(reversed(params))
        off = 0
        self._spans = []
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self._spans.append((p, off, n))
            off += n
        # contiguous buckets of ~equal size
        n_buckets = max(1, min(n_buckets, len(order)))
        target = (total + n_buckets - 1) // n_buckets
        self.buckets = []  # (start, end, [params])
        start, cur, members = 0, 0, []
        for p, o, n in self._spans:
            members.append(p)
            cur = o + n
            if cur - start >= target and len(self.buckets) < n_buckets - 1:
                self.buckets.append((start, cur, members))
                start, members = cur, []
        if members:
            self.buckets.append((start, cur, members))
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        if self._active:
            for bi, (_, _, members) in enumerate(self.buckets):
                for p in members:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self._use_avg = self._active and dist.get_backend(process_group) == 'nccl'
        self.reset()

    def _launch(self, bi):
        s, e, _ = self.buckets[bi]
        op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
        self._handles.append(dist.all_reduce(self.flat[s:e], op=op, group=self.group, async_op=True))

    def _make_hook(self, bi):
        def hook(param):
            self._pending[bi] -= 1
            # RCCL/NCCL pair collectives across ranks by ISSUE ORDER: buckets are therefore launched strictly in
            # index order on every rank, whatever order autograd happens to finish them in (a parameter unused on
            # one rank, a data-dependent branch).  A finished bucket waits until all lower ones have been issued.
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next)
                self._next += 1
        return hook

    def reset(self):
        self._pending = [len(m) for _, _, m in self.buckets]
        self._handles = []
        self._next = 0  # lowest bucket not yet issued

    def zero_grad(self):
        """One memset instead of one per parameter; keeps ``param.grad`` aliased to the flat buffer."""
        self.flat.zero_()
        for p, o, n in self._spans:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + o * self.flat.element_size():
                p.grad = self.flat[o:o + n].view_as(p)
        self.reset()

    def finish(self):
        """Wait for the bucket all-reduces launched during backward (and launch any whose parameters
        received no gradient this step), leaving the averaged gradient in ``param.grad``."""
        if not self._active:
            return
        while self._next < len(self.buckets):  # buckets holding a parameter that got no gradient: same order
            self._launch(self._next)
            self._next += 1
        for h in self._handles:
            h.wait()
        if not self._use_avg:
            self.flat.div_(self.world)
        self.reset()

    def broadcast_parameters(self, src=0):
        if not self._active:
            return
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)


# ------------------------------------------------------------------------------------------------
# Global loss normalisation.  The reference computes the loss on the GATHERED batch
# (trainer/uflow_trainer.py:48-54) and divides by the global sum of the mask (utils/uflow_utils.py:293) or
# its global mean (losses/flow_loss.py:27).  With a sharded batch a rank only sees its own mask sum M_r;
# the loss terms ask `global_denominator(M_r)` for the quantity to divide by instead:
#       off (default)  ->  M_r                      (per-rank normalisation, no communication)
#       on             ->  (sum_r M_r) / world      (one scalar all-reduce, no gradient)
# so that the AVERAGE over ranks of  N_r / (sum_r M_r / world)  is  (sum_r N_r) / (sum_r M_r): the
# gradient all-reduce(AVG) then yields exactly the gradient of the gathered-batch loss.  (Additive epsilons
# such as census_loss's 1e-6 are divided by `world` by the caller through `world_size()`.)
_global_norm_group = None
_global_norm_on = False


def enable_global_loss_norm(on=True, process_group=None):
    global _global_norm_on, _global_norm_group
    _global_norm_on = bool(on) and dist.is_available() and dist.is_initialized()
    _global_norm_group = process_group


def global_loss_norm_enabled():
    return _global_norm_on


def world_size():
    return dist.get_world_size(_global_norm_group) if _global_norm_on else 1


def global_denominator(local_sum):
    """local mask sum (0-d or 1-element tensor, no grad) -> its mean over ranks when global normalisation is on."""
    if not _global_norm_on:
        return local_sum
    t = local_sum.detach().clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_global_norm_group)
    return t / dist.get_world_size(_global_norm_group)
