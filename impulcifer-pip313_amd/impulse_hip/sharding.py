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


def device_shards(n_channels, n_devices, keep_pairs=True):
    """[(device index, lo, hi)] - the non-empty contiguous channel blocks of shard_channels over n_devices, in channel
    order: what ONE process hands its per-device contexts (impulse_hip._native.device_contexts)."""
    out = []
    for r in range(max(1, int(n_devices))):
        lo, hi = shard_channels(n_channels, max(1, int(n_devices)), r, keep_pairs)
        if hi > lo:
            out.append((r, lo, hi))
    return out


def run_sharded(contexts, shards, work):
    """work(context, lo, hi) for every shard, one host thread per device (the calling thread takes the first shard);
    results in shard order.  Each thread runs with its context installed as the package default (using_context)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import _native

    def call(item):
        r, lo, hi = item
        with _native.using_context(contexts[r]):
            return work(contexts[r], lo, hi)

    if len(shards) <= 1:
        return [call(s) for s in shards]
    with ThreadPoolExecutor(max_workers=len(shards) - 1, thread_name_prefix="impulse-dev") as pool:
        rest = [pool.submit(call, s) for s in shards[1:]]
        first = call(shards[0])
        return [first] + [f.result() for f in rest]


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


def _launch_tag(token):
    import hashlib
    return hashlib.sha256(b"impulse_hip rendezvous:" + (token if isinstance(token, bytes) else str(token).encode())).digest()[:16]


def share_unique_id(rank, path, make_id=None, timeout_s=120.0, token=b""):
    """File rendezvous for the 128-byte RCCL unique id: rank 0 creates it (make_id()) and publishes it at ``path``
    (written beside and renamed, so a reader never sees half of it); every other rank waits for the file.

    The file is self-validating: it starts with a 16-byte tag of ``token``, something every rank of ONE launch shares
    and another launch does not (bench.py: run id, port and the launcher's pid), and readers skip files with another
    tag - a rank that starts before rank 0 cannot pick up the id an earlier launch left behind (ncclCommInitRank with a
    mismatched id has no timeout).  A file that already carries THIS launch's tag cannot be told from a stale one:
    rank 0 refuses it loudly instead of guessing."""
    import os
    import time
    tag = _launch_tag(token)
    if rank == 0:
        try:
            with open(path, "rb") as fh:
                old = fh.read(16)
            if old == tag:
                raise FileExistsError(f"{path} already holds a communicator id with this launch's tag: the rendezvous path "
                                      "(or the token) must be private to one launch - remove the file or pick another path")
            os.remove(path)                               # another launch's leftover: readers skip it by its tag anyway
        except FileNotFoundError:
            pass
        uid = make_id()
        if len(uid) != 128:
            raise ValueError("the unique id is 128 bytes")
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as fh:
            fh.write(tag + uid)
        os.replace(tmp, path)
        return uid
    t0 = time.monotonic()
    while True:
        try:
            with open(path, "rb") as fh:
                blob = fh.read()
            if len(blob) == 144 and blob[:16] == tag:
                return blob[16:]
        except FileNotFoundError:
            pass
        if time.monotonic() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no RCCL unique id for this launch at {path} after {timeout_s:.0f} s")
        time.sleep(0.01)


def broadcast_plan_spectrum_rccl(plan, ctx, rank, world_size, unique_id=None, rendezvous_path=None, src=0, token=b""):
    """The spectrum broadcast done by libimpulse_hip itself over RCCL (imp_comm_*): no torch, no mpi4py in the data
    path.  Collective.  The 128-byte communicator id is either given (``unique_id``: made by rank 0 with
    _native.comm_unique_id() and handed round by the launcher's own control plane) or exchanged through a file
    (``rendezvous_path`` + ``token``, see share_unique_id).  Returns the bytes broadcast."""
    import os
    from . import _native
    if unique_id is None:
        if rendezvous_path is None:
            raise ValueError("give the communicator id or a rendezvous path")
        unique_id = share_unique_id(rank, rendezvous_path, _native.comm_unique_id, token=token)
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
