"""Multi-GPU plumbing for the batch-of-streams mode.

Video streams are independent sequential state machines, so the path shards by
stream with NO data-path collective: global stream g lives on rank g % world
(static).  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU
hosts) is used only for the barrier around the timed region, the
max-over-ranks of the elapsed time and the gather of the per-rank throughput
counters - and, when one rank owns ingest/decode (`fan_out`), for the scatter of
the frame payloads to the ranks that own the streams: the one real exchange
step of the batch-of-streams mode (point-to-point over xGMI, no reduction of
image data anywhere).
"""
import os


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def streams_of_rank(rank, world, total_streams):
    """Global stream ids owned by `rank` (round-robin, SURVEY 8e "stream s -> GPU s mod N")."""
    return [g for g in range(total_streams) if g % world == rank]


class Comm:
    """Thin wrapper so single-process runs need no torch at all."""

    def __init__(self, backend=None):
        self.rank, self.local_rank, self.world = env_world()
        self.dist = None
        self.torch = None
        self.device = "cpu"
        # VS_DIST_FORCE=1: a process group at world size 1 as well (exercises the RCCL path on a one-GPU box: tests/test_dist_rccl.py)
        if self.world > 1 or (os.environ.get("VS_DIST_FORCE") == "1" and "RANK" in os.environ):
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            if backend is None:
                # VS_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks
                backend = os.environ.get("VS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = "cuda"
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def device_sync(self):
        if self.torch is not None and self.device == "cuda":
            self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_counters(self, values):
        """all_gather of a small fixed-size counter vector -> list (per rank) of lists."""
        if self.dist is None:
            return [list(values)]
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.device)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[float(x) for x in o.tolist()] for o in out]

    def fan_out(self, payloads, nbytes, src=0):
        """Scatter of frame payloads from the ingest rank: `payloads` (on `src` only) is a list with one
        uint8 numpy array of `nbytes` bytes per rank (the frames of the streams that rank owns); every
        rank gets its share as a uint8 tensor on its device (nccl) or in host memory (gloo).  Returns
        (tensor, seconds): the time is bracketed by a barrier and device syncs on both sides.
        Single process: the payload itself, 0 s."""
        import time
        if self.dist is None:
            import numpy as np
            return np.ascontiguousarray(payloads[0]).reshape(-1), 0.0
        torch = self.torch
        recv = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        chunks = None
        if self.rank == src:
            assert len(payloads) == self.world and all(p.nbytes == nbytes for p in payloads)
            chunks = [torch.from_numpy(p.reshape(-1)).to(self.device) for p in payloads]   # staged on the ingest GPU first
        self.device_sync()
        self.dist.barrier()
        t0 = time.perf_counter()
        self.dist.scatter(recv, chunks, src=src)
        self.device_sync()
        self.dist.barrier()
        return recv, time.perf_counter() - t0

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
