"""Multi-GPU sharding of the optimizer/predictor path.

Frames are independent (fresh predictor and coder state per frame, src/codec.cpp:156-159,240-242),
so the path shards by contiguous frame ranges, one process per GPU, with NO collective on the data
path. The only communication is the final gather of the per-frame results (8 doubles + 4 ints per
frame) to the rank that runs the range coder / writes the file, in frame order.
"""
import numpy as np


def frame_shard(nframes: int, world: int, rank: int):
    """Contiguous, balanced [lo, hi) range of frames for `rank` (the first nframes % world ranks
    get one extra frame). Every frame belongs to exactly one rank; order is preserved."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(nframes, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_frame_rows(local_rows: np.ndarray, nframes: int, world: int, rank: int, group=None):
    """All ranks contribute their shard's rows ([hi-lo, ncol] float64, frame order); every rank gets the
    full [nframes, ncol] table in frame order. Uses one all_gather of equal-sized padded blocks."""
    import torch
    import torch.distributed as dist
    local_rows = np.ascontiguousarray(local_rows, dtype=np.float64)
    ncol = local_rows.shape[1] if local_rows.ndim == 2 else 0
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local_rows.reshape(nframes, ncol)
    per = (nframes + world - 1) // world
    pad = np.zeros((per, ncol), np.float64)
    lo, hi = frame_shard(nframes, world, rank)
    pad[:hi - lo] = local_rows
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(pad).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = np.zeros((nframes, ncol), np.float64)
    for r in range(world):
        l, h = frame_shard(nframes, world, r)
        out[l:h] = parts[r].cpu().numpy()[:h - l]
    return out


# ---------------------------------------------------------------------------------------------
# One frame's candidate population across GPUs (BASELINE config 5)
# ---------------------------------------------------------------------------------------------
class SacoptEvaluator:
    """What optimize_split needs from a device, both through the C ABI of libsacopt:
      get_costs  FrameCoder::GetCost for a slice of candidates (src/codec.cpp:341-352), CostGolomb mean carried in call order;
      recost     the same slice costed again from another start state (sacopt_recost_last): the cascade is not re-run and
                 the residuals never leave the GPU.
    frames: (s0, s1) numpy arrays or (dev_ptr0, dev_ptr1, numsamples)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._shape = None

    def get_costs(self, profile, frames, start, n, coefs, state):
        """coefs[f]: [k, ncoef]. Returns (costs[k, nframes], end state[nframes])."""
        cc = np.stack([np.asarray(c, np.float64) for c in coefs])
        costs, st = self.ctx.get_costs(profile, frames, cc, frm=start, n=n, state=state)
        self._shape = cc.shape[:2]
        return costs.T.copy(), st

    def recost(self, state):
        costs, st = self.ctx.recost_last(self._shape[0], self._shape[1], state)
        return costs.T.copy(), st


def _gather_rows(local, nrows, world, rank, group):
    return gather_frame_rows(np.asarray(local, np.float64).reshape(-1, np.asarray(local).shape[-1]), nrows, world, rank, group)


def _numsamples(fr):
    return int(fr[2]) if len(fr) > 2 else len(fr[0])


def optimize_split(ev, profile, frames, prof, ncoef_loop, world=1, rank=0, group=None, stats=None):
    """FrameCoder::Optimize (src/codec.cpp:354-408) with every round's candidate grid split over `world` ranks.

    Each rank evaluates a contiguous slice of the round's candidates (for all `frames`, same profile table `prof` =
    ncoef x {vmin, vmax, vdef}); costs are all-gathered and every rank takes the same argmin with the reference's strict
    `<` in call order (lowest index wins ties, codec.cpp:399). The one CostGolomb instance of an Optimize call carries its
    running mean from candidate to candidate (codec.cpp:364), i.e. across ranks. It forgets its start (0.98^n), so:
      pass 1  every rank costs its slice from 0 (rank 0: from the true carried state); the mean it ends with is the GUESS
              its successor starts from -- one all-gather of nframes doubles per rank;
      pass 2  every rank costs its slice again from its predecessor's guess (ev.recost: residuals stay on the GPU) and
              reports whether it still ends where pass 1 ended; if every rank whose end feeds a successor does, all costs
              are exact by induction from rank 0 -- one all-gather of the costs + end states + flags;
      else    (windows shorter than ~4k samples) the slices are re-costed rank after rank from the true state, one
              broadcast each. Exact in every case; the guesses only save time.
    Collectives run on device tensors when the backend is nccl (RCCL). stats (optional dict) counts rounds whose guesses
    verified / that needed the chain.
    Returns (best[nframes, ncoef], traces: per frame rows {coef (-1 = baseline call), x, cost} in call order)."""
    import torch
    import torch.distributed as dist
    prof = np.asarray(prof, np.float64).reshape(-1, 3)
    ncoef, nf = len(prof), len(frames)
    if ncoef_loop > ncoef:
        raise ValueError("ncoef_loop exceeds the profile's coefficient count")
    cur = np.tile(prof[:, 2], (nf, 1))            # profile.coefs[i].vdef per frame
    wn = [_numsamples(fr) // 8 for fr in frames]  # codec.cpp:356-359
    st = [(_numsamples(fr) - w) // 2 for fr, w in zip(frames, wn)]
    state = np.zeros(nf)                          # CostGolomb::mean_err, one instance per Optimize call
    best_cost = np.zeros(nf)
    traces = [[] for _ in range(nf)]
    dev = None
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    for i in range(ncoef_loop):
        xs = []
        x, xinc = prof[i, 0], (prof[i, 1] - prof[i, 0]) / 10.
        while x <= prof[i, 1]:                    # codec.cpp:392-403
            xs.append(x)
            x += xinc
        cands = [(-1, prof[0, 2])] * (i == 0) + [(i, x) for x in xs]  # the baseline call (codec.cpp:386) rides with round 0
        nc = len(cands)
        lo, hi = frame_shard(nc, world, rank)
        mine = cands[lo:hi]
        k = len(mine)
        coefs = []
        for f in range(nf):
            cf = np.tile(cur[f], (k, 1))
            for c, (ci, x) in enumerate(mine):
                cf[c, max(ci, 0)] = x
            coefs.append(cf)
        if world == 1:
            costs, state = ev.get_costs(profile, frames, st, wn, coefs, state)
        else:
            # pass 1
            c1, e1 = (ev.get_costs(profile, frames, st, wn, coefs, state if rank == 0 else np.zeros(nf))
                      if k else (np.zeros((0, nf)), np.zeros(nf)))
            guesses = _gather_rows(e1[None, :], world, world, rank, group)
            # pass 2
            c2, e2 = (c1, e1) if (rank == 0 or not k) else ev.recost(guesses[rank - 1])
            ok = 1.0 if e2.tobytes() == e1.tobytes() else 0.0
            per = (nc + world - 1) // world
            blk = np.zeros((per + 1, nf + 1))
            blk[:k, :nf] = c2
            blk[per, :nf], blk[per, nf] = e2, ok
            allb = _gather_rows(blk, world * (per + 1), world, rank, group).reshape(world, per + 1, nf + 1)
            active = min(world, nc)               # ranks with a non-empty slice (empty ones are trailing)
            if all(allb[r, per, nf] == 1.0 for r in range(active - 1)):   # the last slice's end feeds nobody
                costs = np.zeros((nc, nf))
                for r in range(active):
                    l, h = frame_shard(nc, world, r)
                    costs[l:h] = allb[r, :h - l, :nf]
                state = allb[active - 1, per, :nf].copy()
                if stats is not None:
                    stats["verified"] = stats.get("verified", 0) + 1
            else:
                if stats is not None:
                    stats["chained"] = stats.get("chained", 0) + 1
                costs = np.zeros((nc, nf))
                for r in range(active):
                    l, h = frame_shard(nc, world, r)
                    buf = torch.zeros((h - l + 1, nf), dtype=torch.float64)
                    if rank == r:
                        c_r, e_r = ev.recost(state)
                        buf = torch.from_numpy(np.concatenate([c_r, e_r[None, :]]))
                    buf = buf.to(dev)
                    dist.broadcast(buf, src=r if group is None else dist.get_global_rank(group, r), group=group)
                    b = buf.cpu().numpy()
                    costs[l:h] = b[:-1]
                    state = b[-1].copy()
        for f in range(nf):
            best_x = cur[f, i]
            for c, (ci, x) in enumerate(cands):
                traces[f].append((float(ci), x, costs[c, f]))
                if ci < 0:
                    best_cost[f] = costs[c, f]
                elif costs[c, f] < best_cost[f]:  # codec.cpp:399
                    best_x, best_cost[f] = x, costs[c, f]
            cur[f, i] = best_x                    # codec.cpp:406
    return cur, [np.array(t, np.float64).reshape(-1, 3) for t in traces]
