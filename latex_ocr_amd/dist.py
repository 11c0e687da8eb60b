"""Data-parallel plumbing.  DATA PLANE on a GPU: RCCL over xGMI through the C ABI of liblxo.so (lxo_comm_init /
lxo_allreduce_bucket, csrc/comm.hip: ncclAllReduce on a side HIP stream) -- the same entry points a non-PyTorch binding would
use.  CONTROL PLANE: a torch.distributed process group of any backend (gloo suffices) that carries the
128-byte RCCL id from rank 0, host barriers and a few host scalars.  On CPU tensors (the gloo tests, bench.py --sim) and where
two ranks share one GPU (tests/test_gpu_dp2.py: RCCL refuses duplicate devices) the data plane is torch.distributed itself.
One process per GPU; samples are independent through encoder,
decoder and loss, so the only exchanges are (1) the global token count that normalises the
loss (model/img2seq.py:69-71 takes the mean over ALL unmasked tokens of the batch) and (2)
the gradient sum.  Gradients are all-reduced in buckets on a side HIP stream as soon as
backward has finalised them (y_W_o before the recurrence, the rest of the decoder after it,
then one bucket per encoder layer from conv6 down; conv2 + conv1 last), overlapping the
remaining backward; Adam is replicated.

The token count never synchronises the host: it is a sum of host-known integers, so each rank uploads its own count at the START of
the step and all-reduces it on a dedicated stream while the forward runs; the loss kernel reads the global count from device memory
(lxo_ce_loss_fwd_bwd_dev).  With host-ordered buckets (the default on a GPU) finish() DOES block the host once per step, until the
helper thread has enqueued the last bucket, i.e. until the GPU has produced the step's last gradients: the host then runs at most one
step ahead of the device (measured: world-1 step 10.65 ms host-ordered against 10.55 plain).
A failing collective is fatal for the job: the worker keeps draining its queue without issuing further collectives, finish() raises
on this rank, and the caller must tear the process group down (train.py / bench.py do: an uncaught exception ends the process, which
is what makes the peers' pending collectives fail instead of hanging).

How a bucket's all-reduce is ordered behind the kernels that produce it: NOT with hipStreamWaitEvent.  On this runtime
(ROCm 7.2, MI355X) a stream that waits for an event recorded on the compute stream slows the compute stream's own
dependent-launch chains by ~0.25 ms per step (tools/queue_probe.py: one such wait per step is enough, every event flag
behaves the same, the reverse direction -- the compute stream waiting for another stream -- is free).  So the compute
stream only RECORDS an event per bucket; a helper thread waits for it on the host and then enqueues the collective on
the side stream ("host-ordered": LXO_DP_HOST_ORDERED=1; the default on a GPU in rounds 3-5).  Round 6: with the decoder in two persistent launches
the stream-side wait measures FASTER than the helper thread (7.47 vs 7.52 ms at world 1) and is the default again.
"""
import ctypes
import os
import queue
import threading

import torch
import torch.distributed as td

from . import _abi


class LxoComm(object):
    """RCCL communicator behind the C ABI (include/lxo.h: lxo_comm_*).  The id travels over the control-plane process group."""

    def __init__(self, device, rank, world, lib=None):
        self.lib = lib if lib is not None else _abi.load()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)                     # ncclCommInitRank binds the communicator to the current device
        idbuf = ctypes.create_string_buffer(_abi.LXO_COMM_ID_BYTES)
        if rank == 0:
            self._ck(self.lib.lxo_comm_unique_id(idbuf), "lxo_comm_unique_id")
        box = [bytes(idbuf.raw)]
        if world > 1:
            td.broadcast_object_list(box, src=0)
        self.handle = ctypes.c_void_p()
        self._ck(self.lib.lxo_comm_init(box[0], rank, world, ctypes.byref(self.handle)), "lxo_comm_init")
        try:                                                       # librccl prints its version banner through C stdio at the first communicator: out with it now,
            ctypes.CDLL(None).fflush(None)                         # not at process exit behind whatever the caller prints last (bench.py's JSON line, on every rank's pipe)
        except Exception:
            pass
        r, w = ctypes.c_int(-1), ctypes.c_int(-1)
        self._ck(self.lib.lxo_comm_info(self.handle, ctypes.byref(r), ctypes.byref(w)), "lxo_comm_info")
        assert (r.value, w.value) == (rank, world), (r.value, w.value, rank, world)
        self.ranks_seen = w.value

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, (self.lib.lxo_comm_last_error() or b"").decode()))

    _DT = {torch.float32: _abi.LXO_F32, torch.bfloat16: _abi.LXO_BF16, torch.int32: _abi.LXO_I32}

    def all_reduce(self, t, stream, ready_event=None):
        """in-place sum of a contiguous CUDA tensor over ranks, enqueued on `stream` (behind ready_event when given)"""
        assert t.is_cuda and t.is_contiguous() and t.dtype in self._DT, (t.device, t.dtype)
        ev = ctypes.c_void_p(ready_event.cuda_event) if ready_event is not None else ctypes.c_void_p(0)
        self._ck(self.lib.lxo_allreduce_bucket(self.handle, ctypes.c_void_p(t.data_ptr()), t.numel(), self._DT[t.dtype],
                                               ctypes.c_void_p(stream.cuda_stream), ev), "lxo_allreduce_bucket")

    def close(self):
        if self.handle:
            self.lib.lxo_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()


class DataParallel(object):
    def __init__(self, device="cuda:0", comm=None):
        """comm: "abi" = gradient buckets and the token count on RCCL through liblxo's C ABI (default on a GPU; LXO_DP_COMM=torch
        overrides), "torch" = on the torch.distributed process group (CPU tensors; two ranks sharing one GPU)."""
        self.device = torch.device(device)
        self.world = td.get_world_size()
        self.rank = td.get_rank()
        self.cuda = self.device.type == "cuda"
        if comm is None:
            comm = os.environ.get("LXO_DP_COMM", "abi") if self.cuda else "torch"
        self.lxo = LxoComm(self.device, self.rank, self.world) if (self.cuda and comm == "abi") else None
        self.side = torch.cuda.Stream(self.device) if self.cuda else None
        self._pending = False
        # token-count exchange: its own stream (never queued behind gradient buckets), pinned staging ring
        self.cnt_stream = torch.cuda.Stream(self.device) if self.cuda else None
        self._cnt_pin = [torch.zeros(1, dtype=torch.float32).pin_memory() for _ in range(4)] if self.cuda else None
        self._cnt_ev = [None] * 4              # event behind each slot's last H2D copy: the host must not rewrite a slot the copy has not read
        self._cnt_i = 0
        self.grad_dtype = None                 # None = reduce gradients as f32 (34.5 MB); torch.bfloat16 halves the bytes (opt-in, see reduce_range_fn)
        self.exposed_ms = []                   # (start, end) event pairs around finish(): the all-reduce time the compute stream waited for
        # round 6: the default is the stream-side wait (side.wait_event) again.  Host ordering was introduced in round 3 because a stream waiting for a
        # compute-stream event slowed the ~1000 dependent launches of the launch-per-step decoder by 0.25 ms per step; with the decoder in two persistent
        # launches that cost is gone (world 1 on RCCL: 7.46-7.48 ms stream-ordered, 7.50-7.55 host-ordered, 7.37 without the exchange) and the
        # stream-ordered form needs no helper thread and does not stop the host in finish().  LXO_DP_HOST_ORDERED=1 restores the helper thread.
        ho = os.environ.get("LXO_DP_HOST_ORDERED", "0")
        self.host_ordered = (self.cuda and ho == "1") or ho == "force"       # "force": also on CPU tensors (gloo tests of the helper thread)
        self._q = None
        self._err = None
        if self.host_ordered:
            self._q = queue.Queue()
            self._thr = threading.Thread(target=self._bucket_worker, name="lxo-dp-buckets", daemon=True)
            self._thr.start()

    def _sum(self, t, stream=None):
        """in-place sum over ranks on the data plane; `stream` = the stream the collective is enqueued on (RCCL through the ABI)"""
        if self.lxo is not None:
            self.lxo.all_reduce(t, stream if stream is not None else torch.cuda.current_stream(self.device))
        else:
            td.all_reduce(t, op=td.ReduceOp.SUM)

    def _reduce(self, seg):
        st = torch.cuda.current_stream(self.device) if self.cuda else None
        if self.grad_dtype is not None:
            low = seg.to(self.grad_dtype)
            self._sum(low, st)
            seg.copy_(low)
        else:
            self._sum(seg, st)

    def _bucket_worker(self):
        """Host-ordered buckets: wait (on the host) for the event behind a bucket's gradients, then enqueue its all-reduce on the
        side stream.  Collectives are issued by this thread only between the first comm() and finish() of a step, in bucket order,
        so every rank issues them in the same order; the main thread issues its own (token count, barriers) outside that window."""
        if self.cuda:
            torch.cuda.set_device(self.device)
        while True:
            item = self._q.get()
            try:
                if item is None:
                    return
                ev, seg = item
                if self._err is None:
                    if ev is None:                  # CPU tensors: nothing to wait for
                        self._reduce(seg)
                    else:
                        ev.synchronize()
                        with torch.cuda.stream(self.side):
                            self._reduce(seg)
            except BaseException as e:          # surfaced by finish()
                self._err = e
            finally:
                self._q.task_done()

    def close(self):
        if self._q is not None:
            self._q.put(None)
            self._thr.join(timeout=10)
            self._q = None
        if self.lxo is not None:
            if self.cuda:
                torch.cuda.synchronize(self.device)
            self.lxo.close()
            self.lxo = None

    def sum_count_async(self, n_local):
        """-> (device float32 tensor [1] holding the sum of n_local over ranks, event or None).  Asynchronous w.r.t. the
        host and the compute stream: the consumer makes its stream wait for the event (Engine.loss)."""
        if not self.cuda:
            t = torch.tensor([float(n_local)], dtype=torch.float32)
            td.all_reduce(t, op=td.ReduceOp.SUM)
            return t, None
        slot = self._cnt_i % len(self._cnt_pin)
        self._cnt_i += 1
        pin = self._cnt_pin[slot]
        if self._cnt_ev[slot] is not None:     # a non-syncing caller may run more than 4 steps ahead of the device
            self._cnt_ev[slot].synchronize()
        pin[0] = float(n_local)
        # ONE communicator, two streams: the count of step N + 1 must not overtake the last gradient bucket of step N on the device.  The host
        # order is already fixed (finish() joined the bucket queue before this call), RCCL orders the kernels of one communicator itself, and
        # this wait makes the order explicit in the stream graph as well (the count stream waiting for the SIDE stream costs the compute
        # stream nothing -- tools/queue_probe.py: only waits on compute-stream events do)
        if self.side is not None:
            self.cnt_stream.wait_stream(self.side)
        with torch.cuda.stream(self.cnt_stream):
            t = pin.to(self.device, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.cnt_stream)
            self._cnt_ev[slot] = copied
            self._sum(t, self.cnt_stream)
            ev = torch.cuda.Event()
            ev.record(self.cnt_stream)
        return t, ev

    # ---- control plane: host scalars and small statistics.  With the ABI data plane the process group may be gloo (CPU only), so
    # these travel through host memory; they are off the step's critical path (end-of-epoch scores, bench.py's gathers) ----
    def _ctl(self, t, op):
        if self.lxo is None:
            td.all_reduce(t, op=op)
            return t
        h = t.detach().to("cpu")
        if td.get_backend() == "nccl":           # a device-only group: reduce a device copy
            d = h.to(self.device); td.all_reduce(d, op=op); h = d.cpu()
        else:
            td.all_reduce(h, op=op)
        t.copy_(h)
        return t

    def sum_scalar(self, x):
        t = torch.tensor([float(x)], dtype=torch.float64)
        return float(self._ctl(t, td.ReduceOp.SUM).item()) if self.lxo is not None else float(self._ctl(t.to(self.device), td.ReduceOp.SUM).item())

    def broadcast_scalar(self, x, src=0):
        box = [float(x)]
        td.broadcast_object_list(box, src=src)
        return float(box[0])

    def all_reduce(self, t):
        if self.lxo is not None and t.is_cuda:
            torch.cuda.current_stream(self.device).synchronize()
        self._ctl(t, td.ReduceOp.SUM)

    def all_reduce_max(self, t):
        if self.lxo is not None and t.is_cuda:
            torch.cuda.current_stream(self.device).synchronize()
        self._ctl(t, td.ReduceOp.MAX)

    def barrier(self):
        td.barrier()

    def reduce_range_fn(self, flat):
        """-> comm(lo, hi): sum flat[lo:hi] over ranks, asynchronously w.r.t. the compute stream.
        With grad_dtype = torch.bfloat16 (LXO_DP_BF16=1; off by default) a bucket travels as bf16 (17.3 MB per step instead
        of 34.5 MB): rounded once before the sum, summed by the collective in bf16, widened back.  Relative error of a summed
        gradient element <= (world + 1) * 2^-9 of the largest addend; tests/test_dp_gloo.py states the bound it holds."""
        def comm(lo, hi, ready=None):
            seg = flat[lo:hi]
            if not self.cuda:
                if self.host_ordered:
                    self._q.put((None, seg)); self._pending = True
                else:
                    self._reduce(seg)
                return
            # ready: an event the producer has already recorded behind this range's gradients (lxo_encoder_bwd_ready: possibly on its
            # weight-gradient side stream); else the range is final in compute-stream order here
            ev = ready
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
            if self.host_ordered:
                self._q.put((ev, seg))
            else:
                self.side.wait_event(ev)
                with torch.cuda.stream(self.side):
                    self._reduce(seg)
            self._pending = True
        comm.takes_ready = os.environ.get("LXO_DP_READY_EVENTS", "1") != "0"
        return comm

    def finish(self, timed=None):
        """Make the compute stream wait for every bucket before the optimizer reads the gradients.  timed: bracket the wait
        with events on the compute stream (bench.py reports the exposed all-reduce time from them)."""
        timed = getattr(self, "time_finish", False) if timed is None else timed
        if self._pending and self.host_ordered:
            self._q.join()                      # every bucket of this step has been enqueued on the side stream (CPU: reduced)
            if self._err is not None:
                e, self._err = self._err, None
                raise e
            if not self.cuda:
                self._pending = False
        if self.cuda and self._pending:
            cur = torch.cuda.current_stream(self.device)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.side)
                e1.record(cur)
                self.exposed_ms.append((e0, e1))
            else:
                cur.wait_stream(self.side)
            self._pending = False

    def exposed_allreduce_ms(self):
        """Mean milliseconds per step the compute stream spent waiting in finish(timed=True) (call after a synchronize)."""
        if not self.exposed_ms:
            return None
        v = [a.elapsed_time(b) for a, b in self.exposed_ms]
        self.exposed_ms = []
        return sum(v) / len(v)
