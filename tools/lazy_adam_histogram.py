#!/usr/bin/env python3
"""Round-5 review, next 4(a): WHERE the row-lazy Adam's replay goes.  For the rows a training step is about to touch, every
element-step the catch-up kernel (csrc/adam.hip: adam_rows_catchup_kernel) will replay is put in one of three classes, from the
table state right before the step:

  m == 0    the first moment is exactly zero: the update is exactly 0, only v decays           (skipped exactly by the kernel's
            "never touched" test when the WHOLE 4096-column tile is like that)
  settled   |update| is provably below a quarter ulp of p for the rest of the gap: p's bits cannot change, the kernel runs the two
            moment decays only (2 of 7 instructions) -- when all 1,024 elements of a wave are
  live      p really moves: the 7-instruction element-step of the dense kernel, from registers

The per-element class boundaries follow adam_param_settled with the update's decay ratio b1 / sqrt(b2) per step (the bound the
kernel evaluates every 32 steps); the count is what an IDEAL per-element skip could save -- the kernel's decisions are per wave."""
import math

import numpy as np
import torch


@torch.no_grad()
def replay_histogram(table, ids, lr, beta1=0.9, beta2=0.999, eps=1e-8, tag=""):
    """table: a LazyRowEmbedding bound to its optimizer state; ids: the rows of the coming step (device int64)."""
    if table._opt is None or table._last_step is None:
        print(tag + "no optimizer state yet")
        return None
    m_all, v_all, _ = table._opt
    rows = torch.unique(ids[ids >= 0])
    t_now = table.steps_on_device()
    gap = (t_now - table._last_step[rows].long()).clamp(min=0)                  # steps each row has to replay
    p, m, v = table.weight[rows].float(), m_all[rows].float(), v_all[rows].float()
    F = p.shape[1]
    steps = gap[:, None].expand(-1, F).double()
    total = float(steps.sum())
    if total == 0:
        print(tag + "nothing to replay (every row was touched in the previous step)")
        return None
    zero_m = m == 0
    # |update_j| <= 1.01 lr' |m| / (sqrt(v) b2^128 + eps) now, shrinking by r = b1 / sqrt(b2) per step (adam_param_settled's
    # bound with the exact decay in place of its 256-step worst case); settled once below a quarter ulp of p
    bound = 1.01 * lr * m.abs().double() / (v.double().sqrt() + eps)
    quarter_ulp = torch.ldexp(torch.ones_like(p), torch.frexp(p)[1].to(torch.int32) - 1 - 25).double() * (p != 0)
    r = beta1 / math.sqrt(beta2)
    need = torch.where(bound > quarter_ulp, torch.log(quarter_ulp.clamp(min=1e-300) / bound.clamp(min=1e-300)) / math.log(r),
                       torch.zeros_like(bound))
    need = torch.where(quarter_ulp == 0, torch.full_like(need, float("inf")), need)          # p == 0: any update moves it
    live_steps = torch.minimum(need.clamp(min=0), steps)
    live_steps = torch.where(zero_m, torch.zeros_like(live_steps), live_steps)
    live = float(live_steps.sum())
    zero = float(steps[zero_m].sum())
    settled = total - live - zero
    wave_live = float((live_steps.reshape(rows.shape[0], -1, 1024).amax(dim=2) * 1024).sum()) if F % 1024 == 0 else float("nan")
    out = {"rows": int(rows.shape[0]), "F": F, "optimizer_step": t_now, "gap_mean": float(gap.float().mean()), "gap_max": int(gap.max()),
           "element_steps": total, "share_m_zero": zero / total, "share_settled": settled / total, "share_live": live / total,
           "share_live_at_wave_granularity": wave_live / total,
           "p_zero_share_of_elements": float((p == 0).float().mean())}
    print(tag + "step %d: %d rows x %d columns, gap mean %.0f max %d -> %.3g element-steps: m == 0 %.1f %%, settled %.1f %%, LIVE %.1f %% "
          "(%.1f %% when a wave of 1,024 elements replays until its last element settles); p == 0 for %.1f %% of the elements" %
          (t_now, out["rows"], F, out["gap_mean"], out["gap_max"], total, 100 * out["share_m_zero"], 100 * out["share_settled"],
           100 * out["share_live"], 100 * out["share_live_at_wave_granularity"], 100 * out["p_zero_share_of_elements"]), flush=True)
    return out
