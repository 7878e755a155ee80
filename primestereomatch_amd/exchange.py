"""The one exchange step per frame of the sharded hot path, behind one interface for two transports.

  backend "nccl"  RCCL on device tensors, ordered on the current torch stream (one process per GPU over xGMI) - the product.
  backend "gloo"  the same collective staged through page-locked host tensors: device -> host on the current stream, the
                  host waits for that copy, gloo runs the collective on the host tensors, wait() brings the result back
                  host -> device on the current stream.  It exists so that the N > 1 protocol of bench.py (alternating
                  buffers, pending / finish_pending, stripes.assemble, DispSelect_merge(world)) can run with world_size 2
                  where RCCL cannot: two ranks on ONE GPU (RCCL refuses a device that appears twice in a communicator),
                  or no GPU at all (tests/test_dist_gloo.py, CPU tensors standing in for device tensors).

Both return a work object with wait() for async_op=True; wait() of the staged form must be called on the stream / thread that
issued the collective (bench.py does: finish_pending runs inside step()).  At most one collective per device tensor may be in
flight - bench.py's frame pipeline finishes frame i's exchange before it issues frame i+1's.
"""


class _Done:
    def wait(self):
        return True


class _Staged:
    """gloo collective on host tensors in flight; wait() = collective complete + result on the device (enqueued)."""

    def __init__(self, work, dev_out, host_out):
        self.work, self.dev_out, self.host_out = work, dev_out, host_out

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.dev_out is not None:
            self.dev_out.copy_(self.host_out, non_blocking=True)
            self.dev_out = None
        return True


class Exchange:
    def __init__(self, torch, dist, backend: str):
        if backend not in ("nccl", "gloo"):
            raise ValueError(f"Exchange: unknown backend {backend!r}")
        self.torch, self.dist, self.backend = torch, dist, backend
        self._host = {}
        self.collectives = 0        # issued so far (tests / the bench line)

    # ---- staging ----
    def _host_like(self, t, tag):
        key = (tag, t.data_ptr(), t.numel(), t.dtype)
        h = self._host.get(key)
        if h is None:
            pin = t.is_cuda
            h = self.torch.empty(t.numel(), dtype=t.dtype, pin_memory=pin)
            self._host[key] = h
        return h

    def _to_host(self, t, tag):
        h = self._host_like(t, tag)
        h.copy_(t.reshape(-1), non_blocking=True)
        if t.is_cuda:
            # gloo reads host memory: the copy (and every kernel before it on this stream) must have completed.  This also
            # orders the previous frame's host -> device copy out of the receive buffer before gloo overwrites it.
            ev = self.torch.cuda.Event()
            ev.record()
            ev.synchronize()
        return h

    # ---- collectives ----
    def all_gather(self, recv, send, async_op=False):
        """recv[world * n] <- every rank's send[n], rank-major (all_gather_into_tensor)."""
        self.collectives += 1
        if self.backend == "nccl":
            w = self.dist.all_gather_into_tensor(recv, send, async_op=async_op)
            return w if async_op else _Done()
        hs = self._to_host(send, "s")
        hr = self._host_like(recv, "r")
        w = self.dist.all_gather_into_tensor(hr, hs, async_op=True)
        st = _Staged(w, recv.reshape(-1), hr)
        if not async_op:
            st.wait()
        return st

    def all_reduce_min(self, t, async_op=False):
        """t <- elementwise minimum over the ranks (packed WTA keys: min cost, then lowest d)."""
        self.collectives += 1
        if self.backend == "nccl":
            w = self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, async_op=async_op)
            return w if async_op else _Done()
        h = self._to_host(t, "a")
        w = self.dist.all_reduce(h, op=self.dist.ReduceOp.MIN, async_op=True)
        st = _Staged(w, t.reshape(-1), h)
        if not async_op:
            st.wait()
        return st

    def max_float(self, x: float) -> float:
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = self.torch.tensor([x], dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, xs):
        """[[rank 0's values], [rank 1's], ...] on every rank (diagnostics of the bench line: per-rank compute / collective ms)."""
        dev = "cuda" if self.backend == "nccl" else "cpu"
        n = len(xs)
        world = self.dist.get_world_size()
        t = self.torch.tensor(list(xs), dtype=self.torch.float64, device=dev)
        out = self.torch.empty(world * n, dtype=self.torch.float64, device=dev)
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().view(world, n).tolist()

    def barrier(self):
        self.dist.barrier()
