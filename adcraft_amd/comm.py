"""Bringing up the engine's RCCL communicator without PyTorch: rank 0 asks the C ABI for an RCCL unique id (128 bytes) and
hands it to the other ranks of the node through a file; every rank then calls adc_engine_comm_init.

All ranks of a job run on ONE node (one process per GPU), are children of the same launcher process
(`torch.distributed.run`, or bench.py when it starts its own ranks) and are given RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_PORT in the environment.  The rendezvous file is keyed by (MASTER_PORT, launcher pid), so concurrent jobs and
earlier runs cannot collide; MASTER_PORT itself is not touched (torchrun's own store listens there).
"""
import os
import tempfile
import time

ID_BYTES = 128
_bring_ups = 0       # collective bring-ups this process has taken part in (every rank counts alike): part of each rendezvous name,
                     # so two engines brought up back to back (bench.py: cfg4, then cfg5) can never read each other's files


def env_rank_world():
    """(rank, local_rank, world_size) from the launcher's environment; (0, 0, 1) when not launched as a job"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def rendezvous_path(tag="id"):
    port = os.environ.get("MASTER_PORT", "0")
    job = os.environ.get("ADCRAFT_JOB_ID") or f"{os.getppid()}"
    return os.path.join(tempfile.gettempdir(), f"adcraft_comm_{port}_{job}.{tag}")


def exchange_bytes(rank, world_size, make_payload, tag="id", nbytes=ID_BYTES, timeout=300.0):
    """rank 0 publishes make_payload() (bytes of length nbytes) atomically; the other ranks wait for it.  Returns the
    payload on every rank.  The file is removed by rank 0 once every rank has acknowledged."""
    path = rendezvous_path(tag)
    if world_size <= 1:
        return bytes(make_payload())
    if rank == 0:
        payload = bytes(make_payload())
        if len(payload) != nbytes:
            raise ValueError(f"payload must be {nbytes} bytes")
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(payload)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)           # atomic: readers see nothing or all of it
    else:
        payload = None
        deadline = time.monotonic() + timeout
        while payload is None:
            try:
                with open(path, "rb") as f:
                    data = f.read()
                if len(data) == nbytes:
                    payload = data
                    break
            except FileNotFoundError:
                pass
            if time.monotonic() > deadline:
                raise TimeoutError(f"rank {rank}: no communicator id from rank 0 at {path} after {timeout:.0f} s")
            time.sleep(0.01)
        open(f"{path}.ack{rank}", "wb").close()
    if rank == 0:
        deadline = time.monotonic() + timeout
        for r in range(1, world_size):
            while not os.path.exists(f"{path}.ack{r}"):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank 0: rank {r} never picked up the communicator id ({path})")
                time.sleep(0.01)
        for r in range(1, world_size):
            os.remove(f"{path}.ack{r}")
        os.remove(path)
    return payload


def next_bring_up():
    """sequence number of a collective bring-up (call it once per bring-up, on every rank)"""
    global _bring_ups
    _bring_ups += 1
    return _bring_ups


def init_engine_comm(engine, rank=None, world_size=None, timeout=300.0):
    """collective over the job's ranks: gives `engine` its RCCL communicator (no-op for a single rank).
    Returns (rank, world_size)."""
    r, _, w = env_rank_world()
    rank = r if rank is None else int(rank)
    world_size = w if world_size is None else int(world_size)
    if world_size <= 1:
        return rank, world_size
    uid = exchange_bytes(rank, world_size, engine.comm_unique_id, f"id{next_bring_up()}", ID_BYTES, timeout)
    engine.comm_init(uid, rank, world_size)
    return rank, world_size


class FileReducer:
    """A stand-in for the all-reduce when the ranks cannot form an RCCL communicator - ranks sharing ONE GPU (rehearsing
    the N > 1 host path on a one-GPU box: RCCL refuses two ranks on a device) or no GPU at all (the CPU test of the
    launcher).  Selected by ADCRAFT_DIST_BACKEND=file; never used when each rank has its own GPU.

    Protocol: reduction s = every rank writes <dir>/s.<rank> atomically and reads the others'.  A rank that enters
    reduction s + 2 has seen every rank's file of s + 1, which each rank writes only after reading all of s: its own file
    of s is then deleted.  close(): the last rank to arrive removes the directory."""

    def __init__(self, rank, world_size, timeout=300.0):
        self.rank, self.world, self.timeout, self.seq = int(rank), int(world_size), timeout, 0
        self.dir = rendezvous_path(f"reduce{next_bring_up()}.d")
        if self.world > 1:
            os.makedirs(self.dir, exist_ok=True)

    def allreduce(self, vec, op="sum"):
        import numpy as np
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        if self.world <= 1:
            return vec
        self.seq += 1
        mine = os.path.join(self.dir, f"{self.seq}.{self.rank}")
        with open(mine + ".tmp", "wb") as f:
            f.write(vec.tobytes())
        os.replace(mine + ".tmp", mine)
        if self.seq > 2:
            try:
                os.remove(os.path.join(self.dir, f"{self.seq - 2}.{self.rank}"))
            except FileNotFoundError:
                pass
        parts = []
        deadline = time.monotonic() + self.timeout
        for r in range(self.world):
            p = os.path.join(self.dir, f"{self.seq}.{r}")
            while True:
                try:
                    with open(p, "rb") as f:
                        data = f.read()
                    if len(data) == vec.nbytes:
                        break
                except FileNotFoundError:
                    pass
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank}: rank {r} did not contribute to reduction {self.seq}")
                time.sleep(0.002)
            parts.append(np.frombuffer(data, dtype=np.float64))
        return np.max(parts, axis=0) if op == "max" else np.sum(parts, axis=0)

    def close(self):
        if self.world <= 1 or self.dir is None:
            return
        import shutil
        open(os.path.join(self.dir, f"bye.{self.rank}"), "wb").close()
        try:
            if sum(1 for f in os.listdir(self.dir) if f.startswith("bye.")) >= self.world:
                shutil.rmtree(self.dir, ignore_errors=True)         # everybody is past their last read
        except FileNotFoundError:
            pass
        self.dir = None
