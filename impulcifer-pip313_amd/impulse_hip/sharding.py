"""Channel sharding across the GPUs of one node: one process per GPU, no data-path collective.

Every speaker x ear channel is independent; the only shared datum is the prepared inverse-sweep
spectrum, which rank 0 builds (fp64 host FFT) and every other rank receives through ONE broadcast
(RCCL over xGMI when the process group backend is "nccl"; gloo in the CPU tests).
The reference's counterpart is the thread/process pool over channels
(core/parallel_utils.py:97-152, core/parallel_processing.py:84-140).
"""
import numpy as np


def shard_channels(n_channels, world_size, rank, keep_pairs=True):
    """Contiguous block [lo, hi) of channels for ``rank``.  With ``keep_pairs`` the unit is a
    left/right pair, so interleaved stereo frames never straddle two devices."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    unit = 2 if keep_pairs and n_channels % 2 == 0 else 1
    units = n_channels // unit
    base, extra = divmod(units, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo * unit, hi * unit


def broadcast_bytes(buf, dist, src=0):
    """Broadcast a torch uint8 tensor in place from ``src`` (thin wrapper so tests can use gloo)."""
    dist.broadcast(buf, src=src)
    return buf


def broadcast_plan_spectrum(plan, ctx, dist, torch, device, src=0, via_host=False):
    """Rank ``src`` sends its plan's prepared spectrum; the others load it into their (empty) plan.
    ``via_host`` stages the bytes through host memory (gloo rehearsal); the default keeps them on the
    device end to end (RCCL)."""
    dptr, nbytes = plan.spectrum_buffer()
    if via_host:
        import numpy as np
        host = np.empty(nbytes, dtype=np.uint8)
        if dist.get_rank() == src:
            ctx.d2h(host, dptr)
        t = torch.from_numpy(host)
        dist.broadcast(t, src=src)
        if dist.get_rank() != src:
            ctx.h2d(dptr, host)
        return nbytes
    staging = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if dist.get_rank() == src:
        ctx.d2d(staging.data_ptr(), dptr, nbytes)
        ctx.synchronize()
    dist.broadcast(staging, src=src)
    if dist.get_rank() != src:
        torch.cuda.synchronize(device)
        ctx.d2d(dptr, staging.data_ptr(), nbytes)
        ctx.synchronize()
    return nbytes


def share_unique_id(rank, path, make_id=None, timeout_s=120.0):
    """File rendezvous for the 128-byte RCCL unique id: rank 0 creates it (make_id()) and publishes it at ``path``
    (written beside and renamed, so a reader never sees half of it); every other rank waits for the file.  ``path``
    must be private to one launch (bench.py derives it from the launcher's run id and port)."""
    import os
    import time
    if rank == 0:
        try:
            os.remove(path)                               # a stale id left by an earlier launch must not be picked up
        except OSError:
            pass
        uid = make_id()
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as fh:
            fh.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.monotonic()
    while True:
        try:
            with open(path, "rb") as fh:
                uid = fh.read()
            if len(uid) == 128:
                return uid
        except FileNotFoundError:
            pass
        if time.monotonic() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout_s:.0f} s")
        time.sleep(0.01)


def broadcast_plan_spectrum_rccl(plan, ctx, rank, world_size, unique_id=None, rendezvous_path=None, src=0):
    """The spectrum broadcast done by libimpulse_hip itself over RCCL (imp_comm_*): no torch, no mpi4py in the data
    path.  Collective.  The 128-byte communicator id is either given (``unique_id``: made by rank 0 with
    _native.comm_unique_id() and handed round by the launcher's own control plane) or exchanged through a file
    (``rendezvous_path``: must be private to this launch and must not exist beforehand).  Returns the bytes broadcast."""
    import os
    from . import _native
    if unique_id is None:
        if rendezvous_path is None:
            raise ValueError("give the communicator id or a rendezvous path")
        unique_id = share_unique_id(rank, rendezvous_path, _native.comm_unique_id)
    comm = _native.Comm(ctx, unique_id, rank, world_size)
    try:
        nbytes = comm.broadcast_plan_spectrum(plan, root=src)
        broadcast_plan_spectrum_rccl.last_nranks = comm.nranks_seen()      # ranks as RCCL counted them
        return nbytes
    finally:
        comm.close()
        if rank == 0 and rendezvous_path is not None:
            try:
                os.remove(rendezvous_path)
            except OSError:
                pass


def split_evenly(total, parts):
    """Sizes of ``parts`` near-equal chunks of ``total`` items."""
    base, extra = divmod(int(total), int(parts))
    return np.array([base + (1 if i < extra else 0) for i in range(parts)], dtype=np.int64)
