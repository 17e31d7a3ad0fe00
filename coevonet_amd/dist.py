"""Multi-GPU: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).

The hot path shards by population index: rank r plays every game of individuals [lo, hi) of each role;
HoF / elite / stale nets are replicated (a few MB).  The only exchange per generation is one fused all-gather of, per
role and individual, the play_game triple of its last HoF game (3 x fp64) and its distance to the stale agent (stored
as fp64) = 3 * pop * 32 bytes (19 KB at pop 200): latency-bound, so ONE collective, not three.  Every rank then runs the
same deterministic sharing-score / fitness / rank kernels and rebuilds the same offspring from counter-based noise, so
no weight ever crosses xGMI.
"""
from __future__ import annotations

import os
import sys

import torch
import torch.distributed as dist


def allgather_shards(local: torch.Tensor, world: int) -> torch.Tensor:
    """local [R, n_local, C] (this rank's contiguous index range of every role) -> [R, world*n_local, C], rank-major
    along the index axis.  One all_gather_into_tensor (RCCL on device tensors; gloo moves device tensors through the
    host, which is how the N>1 path is rehearsed with several ranks on one GPU)."""
    if world == 1:
        return local
    R, n_local, Cc = local.shape
    dev = local.device
    flat_in = local.contiguous().reshape(-1)
    if dist.get_backend() == "gloo" and flat_in.is_cuda:
        flat_in = flat_in.cpu()
    out = torch.empty(world * flat_in.numel(), dtype=local.dtype, device=flat_in.device)
    dist.all_gather_into_tensor(out, flat_in)  # flat in, flat out: the form both RCCL and gloo accept
    return out.to(dev).reshape(world, R, n_local, Cc).permute(1, 0, 2, 3).reshape(R, world * n_local, Cc)


def choose_backend(requested, env, cuda_available, device_count):
    """-> (backend, ranks_share_a_device).  RCCL needs one device per rank ("Duplicate GPU detected" otherwise): with more
    local ranks than GPUs (a rehearsal on a one-GPU box) the ranks share the device and the gathers go through gloo.
    Only a KNOWN local world size can say so: launchers that do not export LOCAL_WORLD_SIZE (srun, mpirun) may spread
    WORLD_SIZE over several nodes, where comparing it with this node's device count would send every gather through
    the host - they get nccl."""
    lws = env.get("LOCAL_WORLD_SIZE")
    shared = lws is not None and cuda_available and int(lws) > device_count
    # (no GPU at all - a CPU-only or rehearsal host: RCCL cannot initialise there either)
    return requested or env.get("COEVO_DIST_BACKEND") or ("gloo" if (shared or not cuda_available) else "nccl"), shared


class ShardRehearsal:
    """Rank `rank` of `world` WITHOUT the other ranks: the per-GPU half of a population-sharded run on one GPU (bench.py's
    shard legs).  This rank plays, breeds and rebuilds exactly what rank `rank` of a real run would; the all-gather is
    replaced by tiling this rank's own (reward, distance) shard over the other ranks' index ranges - the same pack / unpack
    copies, no collective.  Results are NOT those of the real run (the other shards' fitness is made up); the work is."""

    rehearsal = True

    def __init__(self, rank, world):
        self.rank, self.world, self.local_rank = int(rank), int(world), 0

    def _tile(self, eng):
        n = eng.hi - eng.lo
        for r in range(self.world):
            if r != self.rank:
                eng.last_reward[:, r * n:(r + 1) * n] = eng.last_reward[:, eng.lo:eng.hi]
                eng.dist_all[:, r * n:(r + 1) * n] = eng.dist_all[:, eng.lo:eng.hi]

    gather_ga = gather_ga2 = _tile

    def gather_ga_packed(self, eng):
        """the fused exchange (GAEngine._packed_exchange): every rank's record := this rank's, in ONE copy launch (what the
        collective is on hardware: one kernel)"""
        eng.pack_all.copy_(eng.pack_local.unsqueeze(0).expand_as(eng.pack_all))

    def start_gather_timing(self):
        pass

    def gather_times_us(self):
        return []

    def barrier(self):
        pass

    def max_over_ranks(self, seconds, device):
        return seconds

    def shutdown(self):
        pass


class DistContext:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            backend, shared_device = choose_backend(backend, os.environ, torch.cuda.is_available(),
                                                    torch.cuda.device_count())
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank % max(torch.cuda.device_count(), 1))
            if shared_device:
                # several ranks on ONE device (a rehearsal): their persistent rollout launches would wait for workgroups that
                # the other ranks' launches keep off the chip (coevo_mpe_persistent_fits counts one process) - launches per cycle
                os.environ.setdefault("COEVO_PERSISTENT", "0")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            if self.rank == 0:
                print(f"[coevonet_amd.dist] backend={backend} world={self.world}"
                      + (" (ranks share a device: gathers go through the host, rollouts are launched per env-cycle)"
                         if shared_device else ""),
                      file=sys.stderr, flush=True)
        self._timed = None   # start_gather_timing(): [(start event, end event)] around every gather_* call

    # ---- HIP events around the collectives (bench.py's "allgather_us"): on the caller's stream, so a pair brackets the
    # stream-ordered collective itself (RCCL) or the host-staged copy + gloo exchange (rehearsals)
    def start_gather_timing(self):
        self._timed = []

    def gather_times_us(self):
        """durations of the gather_* calls since start_gather_timing(), after a device synchronize"""
        torch.cuda.synchronize()
        return [1e3 * a.elapsed_time(b) for a, b in (self._timed or [])]

    def _bracket(self, fn, *args):
        if self._timed is None:
            return fn(*args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(*args)
        b.record()
        self._timed.append((a, b))

    def gather_ga(self, eng):
        """fills eng.dist[role][:] and eng.last_reward[:, :, :] for the whole population from every rank's shard"""
        self._bracket(self._gather_ga, eng)

    def gather_ga2(self, eng):
        """the same exchange for the two-role DeepQN engine (dqn_population.DQNGAEngine)"""
        self._bracket(self._gather_ga2, eng)

    def gather_ga_packed(self, eng):
        """the fused form (round 5): eng.pack_local [role][n_local][4] fp64 - written by the rollout's closing launch,
        coevo_mpe_final_step_pack - all-gathered rank-major into eng.pack_all [world][role][n_local][4], which
        coevo_ga_select_gathered reads as it is: ONE collective, no pack / unpack launch around it"""
        self._bracket(self._gather_ga_packed, eng)

    def gather_es(self, eng, what):
        """Co-ES exchanges (evolutionary_strategy.ESEngine.update_device): "stats" = per role and individual the reward
        in the role's slot + the distance to the base net (fp64 pairs, 16 bytes per individual and role); "partials" =
        this rank's chunk partial sums of the update (fp32, chunks/world * P per role), rank-major in eng.partials"""
        self._bracket(self._gather_es, eng, what)

    def _gather_ga(self, eng):
        lo, hi = eng.lo, eng.hi
        local = torch.empty(3, hi - lo, 4, dtype=torch.float64, device=eng.last_reward.device)
        local[:, :, :3] = eng.last_reward[:, lo:hi]
        local[:, :, 3] = eng.dist_all[:, lo:hi]          # fp32 -> fp64 (exact), all roles in one copy
        full = allgather_shards(local, self.world)
        eng.last_reward.copy_(full[:, :, :3])
        eng.dist_all.copy_(full[:, :, 3])               # back to fp32 (exact: they were fp32 values)

    def _gather_ga_packed(self, eng):
        src, out = eng.pack_local.view(-1), eng.pack_all.view(-1)
        if dist.get_backend() == "gloo" and src.is_cuda:   # (a rehearsal of several ranks on one GPU: through the host)
            host_out = torch.empty(out.numel(), dtype=out.dtype)
            dist.all_gather_into_tensor(host_out, src.cpu())
            out.copy_(host_out)
            return
        dist.all_gather_into_tensor(out, src)

    def _gather_ga2(self, eng):
        lo, hi = eng.lo, eng.hi
        local = torch.empty(2, hi - lo, 4, dtype=torch.float64, device=eng.last_reward.device)
        local[:, :, :3] = eng.last_reward[:, lo:hi]
        local[:, :, 3] = eng.dist_all[:, lo:hi]
        full = allgather_shards(local, self.world)
        eng.last_reward.copy_(full[:, :, :3])
        eng.dist_all.copy_(full[:, :, 3])

    def _gather_es(self, eng, what):
        if what == "stats":
            full = allgather_shards(eng.stats[:, eng.lo:eng.hi].contiguous(), self.world)
            eng.stats.copy_(full)
        else:
            blk = eng.part_block
            mine = eng.partials[eng.rank * blk:(eng.rank + 1) * blk]
            src = mine.cpu() if dist.get_backend() == "gloo" else mine.clone()
            out = torch.empty(self.world * blk, dtype=src.dtype, device=src.device)
            dist.all_gather_into_tensor(out, src)
            eng.partials.copy_(out)

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def max_over_ranks(self, seconds: float, device) -> float:
        if self.world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def shutdown(self):
        if self.world > 1 and dist.is_initialized():
            dist.destroy_process_group()
