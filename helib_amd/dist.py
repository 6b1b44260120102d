"""Multi-GPU harness for the one axis this path shards on: independent ciphertexts.

One process per GPU; each rank owns a contiguous slice of the ciphertext batch and a replica
of the (small) key-switch matrix.  There is no data-path collective: torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests) is used only for the
barrier around the timed region and the max-over-ranks of the elapsed time."""
import os


def env_world():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard(total, world, rank):
    """Contiguous block partition of `total` independent ciphertexts: (start, count)."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


class Group:
    def __init__(self, backend=None, device=None):
        self.world, self.rank, self.local_rank = env_world()
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend or "gloo", rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
        self.device = device

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        dev = self.device if self.device is not None else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        dev = self.device if self.device is not None else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
