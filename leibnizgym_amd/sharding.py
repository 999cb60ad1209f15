"""Env sharding over the GPUs of one node.

Environments are independent: GPU g of G owns the contiguous range [g*N/G, (g+1)*N/G) and steps it with its own
handle; there is NO data-path collective.  The in-kernel RNG is keyed by global env id and the reward schedules
by the global env-step count, so the union of the shards is bit-identical to one big engine.

The only exchange is optional: an all-reduce(sum) of the per-step episode statistics (11 floats), used for
logging.  It runs through `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests), every `every` steps, on a side stream so that it never sits on the step's critical path.
"""
import torch

from . import _capi as capi


def shard_range(total: int, rank: int, world: int):
    """(offset, count) of `rank`'s contiguous env range; the first `total % world` ranks get one more."""
    base, rem = divmod(int(total), int(world))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


_MEAN_SLOTS = (0, 1, 2, 3, 4, 5, capi.INFO_SUCCESS_MEAN)     # slots that are means over envs (the rest are counts)


class EpisodeStatsReducer:
    """All-reduce of the info scalars across shards (the north star's "optional episode-stats all-gather")."""

    def __init__(self, engine, global_num_envs: int, group=None, every: int = 1):
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.global_n = float(global_num_envs)
        self.local_n = float(engine.num_envs)
        self.group = group
        self.every = max(1, int(every))
        self.buf = torch.zeros(capi.TF_NUM_INFO, dtype=torch.float32, device=engine.device)
        self._scale_in = torch.ones(capi.TF_NUM_INFO, dtype=torch.float32, device=engine.device)
        self._scale_out = torch.ones(capi.TF_NUM_INFO, dtype=torch.float32, device=engine.device)
        for s in _MEAN_SLOTS:
            self._scale_in[s] = self.local_n
            self._scale_out[s] = 1.0 / self.global_n
        self._side = torch.cuda.Stream(device=engine.device) if engine.device.type == "cuda" else None
        self._count = 0
        self._work = None

    def step(self):
        """Call after engine.step(); returns True when a reduction was launched this step."""
        self._count += 1
        if self._count % self.every:
            return False
        if self._side is not None:
            # snapshot `info` on the MAIN stream: the next engine.step() (queued behind it on that stream) overwrites
            # info[], so the read must not be left to the side stream; only the collective runs there
            main = torch.cuda.current_stream(self.engine.device)
            if self._work is not None:          # a previous reduction still owns `buf`
                self._work.wait()
                main.wait_stream(self._side)
            torch.mul(self.engine.info, self._scale_in, out=self.buf)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                self._work = self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            torch.mul(self.engine.info, self._scale_in, out=self.buf)
            self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.SUM, group=self.group)
        return True

    def result(self) -> torch.Tensor:
        """Global statistics of the last reduction (means over ALL envs of the job, counts summed)."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._side is not None:
            torch.cuda.current_stream(self.engine.device).wait_stream(self._side)
        return self.buf * self._scale_out
