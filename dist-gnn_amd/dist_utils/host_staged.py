"""Rehearsal transport: gloo collectives for GPU tensors, staged through host memory by hand.

``DGNN_GLOO_CUDA=1`` with backend 'gloo' keeps every tensor on the GPU, so several ranks can share one card
and the whole N > 1 flow -- HIP kernels, wire formats, buffers, the self-tuning pipeline -- runs on a one-GPU box.
gloo's own CUDA path crawls and stalls on this image (tools/gloo_cuda_probe.py), its CPU path is sound, so
``install()`` wraps the torch.distributed calls this package makes: a GPU tensor is copied to the host, the
collective runs on the host copy, and ``wait()`` copies the result back. Posting synchronises the current
stream (the copy out), so nothing overlaps: this checks the flow, it is not a transport to measure.
Never installed for 'nccl'.
"""
import torch
import torch.distributed as dist

_real = {}


class _Staged:
    """Stands in for the Work handle of an async collective: wait() finishes the host side and copies back."""

    def __init__(self, finish):
        self._finish = finish

    def wait(self):
        if self._finish is not None:
            fin, self._finish = self._finish, None
            fin()
        return True

    def is_completed(self):
        return self._finish is None


def _ret(finish, async_op):
    if async_op:
        return _Staged(finish)
    finish()
    return None


def _broadcast(tensor, src=None, group=None, async_op=False, **kw):
    if not tensor.is_cuda:
        return _real['broadcast'](tensor, src=src, group=group, async_op=async_op, **kw)
    host = tensor.cpu()
    work = _real['broadcast'](host, src=src, group=group, async_op=True, **kw)

    def finish():
        work.wait()
        if dist.get_rank() != src:
            tensor.copy_(host)
    return _ret(finish, async_op)


def _all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
    if not tensor.is_cuda:
        return _real['all_reduce'](tensor, op=op, group=group, async_op=async_op)
    host = tensor.cpu()
    work = _real['all_reduce'](host, op=op, group=group, async_op=True)

    def finish():
        work.wait()
        tensor.copy_(host)
    return _ret(finish, async_op)


def _all_gather_into_tensor(output_tensor, input_tensor, group=None, async_op=False):
    if not output_tensor.is_cuda:
        return _real['all_gather_into_tensor'](output_tensor, input_tensor, group=group, async_op=async_op)
    hin = input_tensor.cpu().contiguous()
    hout = torch.empty(output_tensor.shape, dtype=output_tensor.dtype)
    work = _real['all_gather_into_tensor'](hout, hin, group=group, async_op=True)

    def finish():
        work.wait()
        output_tensor.copy_(hout)
    return _ret(finish, async_op)


def _all_gather(tensor_list, tensor, group=None, async_op=False):
    if not tensor.is_cuda:
        return _real['all_gather'](tensor_list, tensor, group=group, async_op=async_op)
    hosts = [torch.empty(t.shape, dtype=t.dtype) for t in tensor_list]
    work = _real['all_gather'](hosts, tensor.cpu(), group=group, async_op=True)

    def finish():
        work.wait()
        for t, h in zip(tensor_list, hosts):
            t.copy_(h)
    return _ret(finish, async_op)


def _all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
    if not output.is_cuda:
        return _real['all_to_all_single'](output, input, output_split_sizes, input_split_sizes, group=group, async_op=async_op)
    hout = torch.empty(output.shape, dtype=output.dtype)
    work = _real['all_to_all_single'](hout, input.cpu(), output_split_sizes, input_split_sizes, group=group, async_op=True)

    def finish():
        work.wait()
        output.copy_(hout)
    return _ret(finish, async_op)


def _batch_isend_irecv(p2p_op_list):
    if not any(op.tensor.is_cuda for op in p2p_op_list):
        return _real['batch_isend_irecv'](p2p_op_list)
    staged, back = [], []
    for op in p2p_op_list:
        if op.op in (dist.isend, _real.get('isend')):
            host = op.tensor.cpu()
        else:
            host = torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
            back.append((op.tensor, host))
        staged.append(dist.P2POp(op.op, host, op.peer, op.group, op.tag))
    works = _real['batch_isend_irecv'](staged)

    def finish():
        for w in works:
            w.wait()
        for t, h in back:
            t.copy_(h)
    return [_Staged(finish)]


def install():
    """Idempotent. Patches the module attributes, which is how this package calls them (``dist.broadcast(...)``)."""
    if _real:
        return
    for name, fn in (('broadcast', _broadcast), ('all_reduce', _all_reduce),
                     ('all_gather_into_tensor', _all_gather_into_tensor), ('all_gather', _all_gather),
                     ('all_to_all_single', _all_to_all_single), ('batch_isend_irecv', _batch_isend_irecv)):
        _real[name] = getattr(dist, name)
        setattr(dist, name, fn)
