"""One-shot gradient exchange between the per-GPU processes of a node (include/dmf.h "dmf_xgmi_*").

Each rank owns two uncached device buffers (the inbox of tagged words — value + sequence number in one 8-byte store — and
a status block) and maps every peer's pair through HIP IPC; the 64-byte handles travel over the torch.distributed group that
already exists for the job.
`create()` verifies the mapping with a known-answer all-reduce against the group's own all_reduce before the
communicator is handed out; if any rank cannot set it up (IPC refused, a wait timed out, a sum differs) every
rank gets None and the caller stays on the RCCL all-reduce path.
"""
import sys

import torch
import torch.distributed as dist

from . import lib


class Communicator:
    def __init__(self, capacity, group=None, timeout_ms=20000):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if not 2 <= self.world <= 16:
            raise lib.DmfError('xgmi exchange is built for 2..16 ranks, got %d' % self.world)
        self.capacity = int(capacity)
        self._own, self._mapped = [], []
        data_bytes, flag_bytes = lib.xgmi_sizes(self.capacity, self.world)
        data = lib.xgmi_alloc(data_bytes); self._own.append(data)
        flags = lib.xgmi_alloc(flag_bytes); self._own.append(flags)
        mine = (lib.xgmi_export(data), lib.xgmi_export(flags))
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        self.c = lib.XgmiComm(world=self.world, rank=self.rank, capacity=self.capacity, timeout_ms=int(timeout_ms),
                              seq_bias=0)
        for r, (hd, hf) in enumerate(handles):
            if r == self.rank:
                self.c.data[r], self.c.flags[r] = data, flags
            else:
                pd = lib.xgmi_open(hd); self._mapped.append(pd)
                pf = lib.xgmi_open(hf); self._mapped.append(pf)
                self.c.data[r], self.c.flags[r] = pd, pf
        self._seq = 0

    def allreduce_(self, buf):
        """In-place sum over ranks (rank order) of a contiguous fp32 device tensor with numel <= capacity."""
        self._seq += 1
        lib.xgmi_allreduce(self.c, buf, buf.numel(), self._seq)
        return buf

    def rewind(self, steps):
        """The host moved the device step count back by `steps` (graph warm-up): keep sequence numbers rising."""
        self.c.seq_bias += int(steps)

    def status(self):
        """0 = every wait so far was satisfied; 1 = a rank gave up waiting (host sync)."""
        return lib.xgmi_status(self.c)

    def close(self):
        torch.cuda.synchronize()
        for p in self._mapped:
            lib.xgmi_close(p)
        for p in self._own:
            lib.xgmi_free(p)
        self._mapped, self._own = [], []


def cu_share_stream(rank, world, device=None):
    """A stream whose kernels run only on rank's contiguous share of the GPU's compute units (hipExtStreamCreateWithCUMask).

    For REHEARSALS of the exchange with several ranks on ONE GPU (tests/test_gpu_dp.py, `DMF_CU_SHARE=1 bench.py --gpus N`
    on a one-GPU box) — never needed on a node, where a rank's GPU holds its own kernels only.  Why it is needed there
    (measured in round 3 with the three processes of the test): the blocks of the reduce launch WAIT inside the exchange, one
    of them fits a compute unit beside nothing else of its kind (5 waves, 135 registers), and the patch kernel needs a
    compute unit to itself.  The two ranks that reach the exchange first cover all 256 units with waiting blocks
    (157 + 99 in the test), the third rank's kernels find no unit, its flags never come and everybody times out: a dead-lock
    of residency, not of the protocol (with two ranks the second always finds free units).  On its own share of the units a
    rank's blocks are dispatched in order whatever the peers do, and all ranks work through the same block indices."""
    import ctypes
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    hip = lib.hip_runtime_of_torch()                 # (the runtime torch has mapped, not a second copy by bare name)
    if hip is None:
        raise lib.DmfError('libamdhip64 is not mapped into this process')
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    lo, hi = rank * n_cu // world, (rank + 1) * n_cu // world
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(lo, hi):
        mask[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0 or not st.value:
        raise lib.DmfError('hipExtStreamCreateWithCUMask failed (%d)' % rc)
    return torch.cuda.ExternalStream(st.value, device=dev)


def _all_agree(ok, group, device):
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())


ADMISSION_ROUNDS = 6


def create(capacity, group=None, timeout_ms=20000, verbose=True):
    """Build a communicator and prove it before it is handed out; None (on every rank) if that fails.

    The proof is ADMISSION_ROUNDS exchanges whose lengths change from round to round: every parity slot of the inbox is
    REUSED at least twice (round 1's wrong 3-rank sums appeared exactly at the first reuse, in the third exchange — two rounds
    would never have seen them), on seeded random data as well as on a known-answer pattern, each compared bit for bit with
    the rank-ordered sum formed from the group's own all_gather.  Any mismatch or time-out on any rank fails all ranks."""
    backend = dist.get_backend(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    comm, err = None, ''
    try:
        comm = Communicator(capacity, group, timeout_ms=min(timeout_ms, 5000))
    except Exception as e:                      # noqa: BLE001 — any failure means "use RCCL", reported below
        err = '%s: %s' % (type(e).__name__, e)
    ok = _all_agree(comm is not None, group, dev)
    if ok:
        world, rank = comm.world, comm.rank
        cap = comm.capacity
        for rnd in range(ADMISSION_ROUNDS):
            n = max(1, min(cap, cap - (rnd * 977) % max(cap // 2, 1)))      # full length first, then shorter and changing
            if rnd % 2 == 0:                    # known answer
                mine = (torch.arange(n, dtype=torch.float32, device='cuda') % 97) * (rank + 1) + rnd
            else:                               # seeded random data, different on every rank
                gen = torch.Generator(device='cuda').manual_seed(1000 * rnd + rank)
                mine = torch.randn(n, device='cuda', generator=gen)
            parts = [torch.empty(n, device=dev) for _ in range(world)]
            dist.all_gather(parts, mine.to(dev), group=group)
            want = parts[0].to('cuda').clone()
            for r in range(1, world):
                want += parts[r].to('cuda')     # rank order, like the kernel
            buf = mine.clone()
            comm.allreduce_(buf)
            good = comm.status() == 0 and torch.equal(buf, want)
            ok = _all_agree(good, group, dev) and ok
            if not good:
                err = err or 'admission exchange %d of %d failed (status %d, %d of %d elements differ)' % (
                    rnd + 1, ADMISSION_ROUNDS, comm.status(), int((buf != want).sum()), n)
            if not ok:
                break
        comm.c.timeout_ms = int(timeout_ms)
    if not ok:
        if comm is not None:
            try:
                comm.close()
            except Exception:                   # noqa: BLE001
                pass
        if verbose:
            print('[dmf.xgmi] rank %d: one-shot exchange unavailable (%s); using the RCCL all-reduce'
                  % (dist.get_rank(group), err or 'a peer failed'), file=sys.stderr, flush=True)
        return None
    return comm
