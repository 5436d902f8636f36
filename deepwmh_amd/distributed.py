"""Multi-GPU modes of the sliding-window path (one process per GPU, torch.distributed; backend "nccl" = RCCL).

The reference is single-GPU (`CUDA_VISIBLE_DEVICES=<one id>`, /root/reference/deepwmh/main/predict.py:150), so both
modes are new design:

* volume sharding  -- cases are independent: rank r takes cases r, r+W, ...  No data-path collective.
* patch sharding   -- one volume, rank r computes patches r, r+W, ... (all 8 mirrored forwards of a patch stay on
  its rank), producing Gaussian-weighted softmax tiles; ONE exchange step (all-gather of the tiles over xGMI); every
  rank then accumulates all tiles in nnU-Net's x->y->z patch order, so the result is bit-identical to the
  single-GPU path (the accumulation adds rounded products, never fused: see seghead_softmax_accum_kernel).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_cases(cases, rank, world):
    """Round-robin assignment of independent cases to ranks."""
    return list(cases)[rank::world]


def average_gradients(flat_grads, group=None, bucket_elems=None):
    """Data-parallel gradient exchange (config 5; new design, the reference is single-GPU): all-reduce (sum) of the
    flat gradient arena and division by the world size.  With `bucket_elems` the arena is reduced in buckets issued
    back to front (the decoder's large gradients are complete first), each an async RCCL operation."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat_grads
    W = dist.get_world_size(group)
    if not bucket_elems or bucket_elems >= flat_grads.numel():
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    else:
        works, n = [], flat_grads.numel()
        for end in range(n, 0, -bucket_elems):
            works.append(dist.all_reduce(flat_grads[max(0, end - bucket_elems):end], op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in works:
            w.wait()
    flat_grads.div_(W)
    return flat_grads


class OverlappedGradientAverager:
    """The same exchange as `average_gradients`, started while the backward pass is still running.

    The backward pass finishes the parameter groups one after another (loss heads and the full-resolution decoder first,
    then the deep stages that hold most of the 31 M parameters, the first encoder layers last).  `ready(names)` is called
    as soon as the gradients of those parameters are final on the current stream; completed neighbours in the flat arena
    are merged, and every merged run of at least `min_elems` elements is handed to an asynchronous all-reduce (RCCL runs
    it on its own stream, ordered after the kernels queued so far, next to the rest of the backward pass).  `finish()`
    sends what is left, waits, checks that every parameter was sent exactly once, and divides by the world size.
    Reduction results do not depend on how the arena was cut (an element-wise sum over ranks)."""

    def __init__(self, flat_grads, ranges, group=None, min_elems=1 << 20, force_collective=False):
        """ranges: {parameter name: (start, end)} element ranges in flat_grads (disjoint; may include alignment padding).
        force_collective: issue the all-reduces at world size 1 too (a sum over one rank: the identity) -- the single-GPU rehearsal of
        the multi-GPU run (tests/test_gpu_rccl.py): RCCL's kernels then run beside the persistent convolution kernels of the backward
        pass exactly as they will on a node."""
        self.flat, self.ranges, self.group, self.min_elems = flat_grads, dict(ranges), group, int(min_elems)
        self.active = dist.is_initialized() and (dist.get_world_size(group) > 1 or bool(force_collective))
        self.world = dist.get_world_size(group) if self.active else 1
        self.begin()

    def begin(self):
        self.seen, self.pending, self.works, self.sent = set(), [], [], 0

    def _issue(self, lo, hi):
        self.sent += hi - lo
        if self.active:
            self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def ready(self, names):
        for n in names:
            if n in self.seen:
                raise RuntimeError('gradient of %s reported twice in one backward pass' % n)
            self.seen.add(n)
            self.pending.append(tuple(self.ranges[n]))
        self.pending.sort()
        merged = []
        for lo, hi in self.pending:
            if merged and merged[-1][1] == lo:
                merged[-1] = (merged[-1][0], hi)
            else:
                merged.append((lo, hi))
        self.pending = []
        for lo, hi in merged:
            if hi - lo >= self.min_elems:
                self._issue(lo, hi)
            else:
                self.pending.append((lo, hi))

    def finish(self):
        missing = [n for n in self.ranges if n not in self.seen]
        if missing:
            raise RuntimeError('backward pass finished without the gradients of: %s' % ', '.join(sorted(missing)[:8]))
        for lo, hi in self.pending:
            self._issue(lo, hi)
        self.pending = []
        assert self.sent == sum(hi - lo for lo, hi in self.ranges.values())
        for w in self.works:
            w.wait()
        self.last_messages = len(self.works)
        if self.active and self.world > 1:
            self.flat.div_(self.world)
        self.begin()
        return self.flat


class PatchShardedPredictor:
    """Host logic of the patch-sharded mode, independent of where the arithmetic runs.

    backend object (GPU: `HipTileBackend`; CPU tests: an oracle-backed stand-in) provides
      geometry(vshape) -> (A, pad_lo, origins, gauss_or_None)
      tile(vol, origin) -> tensor [K, P0, P1, P2]  (mirror-averaged softmax x Gaussian for that patch)
      zeros(shape) / accumulate(agg, tile, origin) / count_map(A, origins) / finalize(agg, cnt, pad_lo, V)
    """

    def __init__(self, backend, group=None, force_collective=False):
        """force_collective: at world size 1 still go through the exchange step (all_gather_into_tensor over the one rank, asynchronous,
        alternating buffers) instead of the serial shortcut: the single-GPU rehearsal of configs[3] (tests/test_gpu_rccl.py)."""
        self.b = backend
        self.group = group
        self.force = bool(force_collective) and dist.is_initialized()
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.stall_marks = None            # a list: bench.py's configs[3] leg asks how long the accumulate stream stood waiting for the exchange

    def _mark(self, like):
        """A time mark on the stream the tiles are computed on (HIP event) or on the host (CPU stand-ins: gloo's wait blocks the host)."""
        if like.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def exposed_exchange_ms(self):
        """Sum over the marked waits of (stream released by the exchange) - (stream arrived at the wait): the part of the all-gathers
        that the next round's forwards did not cover.  Call after a device synchronisation; clears the marks."""
        marks, self.stall_marks = self.stall_marks or [], []
        return sum((a.elapsed_time(b) if hasattr(a, 'elapsed_time') else 1e3 * (b - a)) for a, b in marks)

    def _exchange_buffers(self, W, K, P):
        """Two send tiles and two gather buffers (W x 16.8 MB each at the BASELINE plan), allocated once per geometry."""
        key = (W, K, tuple(P))
        if getattr(self, '_xkey', None) != key:
            self._xkey = key
            self._send = [self.b.zeros((K,) + tuple(P)) for _ in range(2)]
            self._recv = [self.b.zeros((W, K) + tuple(P)) for _ in range(2)]
        return self._send, self._recv

    def predict(self, vol):
        V = tuple(int(v) for v in vol.shape[1:])
        A, pad_lo, origins, K, P = self.b.geometry(V)
        W, r = self.world, self.rank
        rounds = (len(origins) + W - 1) // W
        agg = self.b.zeros((K,) + A)
        collective = W > 1 or self.force
        send, recv = self._exchange_buffers(W, K, P) if collective else (None, None)

        def drain(item):                                     # round rd's tiles, in the fixed global patch order
            work, gathered, rd = item
            if work is not None:
                a = self._mark(gathered) if self.stall_marks is not None else None
                work.wait()
                if a is not None:
                    self.stall_marks.append((a, self._mark(gathered)))
            for w in range(W):
                j = rd * W + w
                if j < len(origins):
                    self.b.accumulate(agg, gathered[w], origins[j])

        pending = None
        for rd in range(rounds):
            idx = rd * W + r
            if collective:
                # this round's forwards run while the previous round's all-gather is still on the wire; the exchange buffers
                # alternate (round rd-2, the last user of this pair, was drained one iteration ago)
                mine = send[rd & 1]
                if idx < len(origins):
                    mine.copy_(self.b.tile(vol, origins[idx]))
                else:
                    mine.zero_()                             # ragged last round: a zero tile that nobody accumulates
                gathered = recv[rd & 1]
                # the one exchange step (output viewed as the concatenation along dim 0: accepted by RCCL and gloo)
                if mine.is_cuda and dist.get_backend(self.group) == 'gloo':
                    # gloo has no all-gather of device tensors: staged through the host.  Not a production path (RCCL is): it lets W ranks
                    # share ONE GPU, which is how the HIP arithmetic is tested under W > 1 where no multi-GPU node exists (tests/test_gpu_two_ranks.py)
                    torch.cuda.current_stream().synchronize()
                    host = torch.empty((W * K,) + tuple(P), dtype=mine.dtype)
                    dist.all_gather_into_tensor(host, mine.cpu(), group=self.group)
                    gathered.view((W * K,) + tuple(P)).copy_(host)
                    work = None
                else:
                    work = dist.all_gather_into_tensor(gathered.view((W * K,) + tuple(P)), mine, group=self.group, async_op=True)
            else:                                            # one rank: nothing to overlap, and the backend may reuse its tile buffer
                drain((None, self.b.tile(vol, origins[idx])[None], rd))
                continue
            if pending is not None:
                drain(pending)
            pending = (work, gathered, rd)
        if pending is not None:
            drain(pending)
        cnt = self.b.count_map(A, origins)
        return self.b.finalize(agg, cnt, pad_lo, V)


class HipTileBackend:
    """The arithmetic of PatchShardedPredictor on the HIP kernels."""

    def __init__(self, predictor):
        from . import _lib
        self.p = predictor                 # deepwmh_amd.sliding_window.SlidingWindowPredictor
        self._lib = _lib
        self.lib = predictor.lib
        self.dev = predictor.device

    def geometry(self, V):
        A, pad_lo, origins = self.p._geometry(V)
        self._V, self._pad_lo = V, pad_lo
        self._gauss = self.p.gaussian() if (self.p.use_gaussian and len(origins) > 1) else None
        return A, pad_lo, origins, self.p.engine.plan.num_classes, self.p.patch

    def zeros(self, shape):
        return torch.zeros(shape, dtype=torch.float32, device=self.dev)

    def tile(self, vol, origin):
        L, lib, p = self._lib, self.lib, self.p
        P, V, K = p.patch, self._V, p.engine.plan.num_classes
        n = 8 if p.do_mirroring else 1
        flips = list(range(8)) if p.do_mirroring else [0]
        patches = p._scratch('patches', (n, vol.shape[0]) + P)
        L.check(lib.wmh_extract_patches(L.ptr(vol), L.ptr(patches), vol.shape[0], V[0], V[1], V[2], L.iarr(self._pad_lo),
                                        P[0], P[1], P[2], n, L.iarr(list(origin) * n), L.iarr(flips), p._stream()), 'extract')
        y, ss, seg_w = p.engine.forward_features(patches)[-1]
        tile = self.zeros((K,) + P)        # 0 + x == x exactly: the tile is the weighted softmax itself
        from .network import LRELU_SLOPE
        L.check(lib.wmh_seghead_softmax_accumulate(L.ptr(y), L.ptr(ss), L.ptr(seg_w), K, L.ptr(self._gauss), L.ptr(tile),
                                                   y.shape[1], P[0], P[1], P[2], P[0], P[1], P[2], L.iarr((0, 0, 0)), n,
                                                   L.iarr(flips), LRELU_SLOPE, p._stream()), 'seghead')
        return tile

    def accumulate(self, agg, tile, origin):
        L, P = self._lib, self.p.patch
        A = tuple(agg.shape[1:])
        for k in range(agg.shape[0]):      # agg[k, window] += tile[k]
            L.check(self.lib.wmh_accumulate_weight(L.ptr(tile[k]), L.ptr(agg[k]), P[0], P[1], P[2], A[0], A[1], A[2],
                                                   L.iarr(origin), self.p._stream()), 'accumulate')

    def count_map(self, A, origins):
        return self.p._count_map(A, origins, self._gauss)

    def finalize(self, agg, cnt, pad_lo, V):
        L = self._lib
        K = agg.shape[0]
        A = tuple(agg.shape[1:])
        seg = torch.empty(V, dtype=torch.uint8, device=self.dev)
        prob = torch.empty((K,) + V, dtype=torch.float32, device=self.dev)
        L.check(self.lib.wmh_finalize_argmax(L.ptr(agg), L.ptr(cnt), L.ptr(prob), L.ptr(seg), K, A[0], A[1], A[2],
                                             L.iarr(pad_lo), V[0], V[1], V[2], self.p._stream()), 'finalize')
        return seg, prob
