"""Projection-mode ("panel") deflation over several ranks -- shared by posComponents (vertex rows) and
constraintsComponents 'pca_blocks' with p = 1 (constraint rows)."""
import numpy as np


def deflate_panels_multirank(eng, comm, n_rows, K):
    """Projection-mode deflation over several ranks.  Per PANEL (up to 16
    components): two histogram all-reduces fix the global threshold, each rank rebuilds the
    exact residual rows of its own candidates, ONE padded all-gather replicates the ~1000
    candidate rows on every rank, every rank runs the identical greedy steps on them (no
    per-component collective), then projects its own shard.  Ranks stay in lock-step because
    every decision is taken on all-reduced / all-gathered data."""
    dev = comm.exchange_device(eng.device_exchange)
    torch = comm._torch
    cap, rl = eng.panel_capacity(), eng.panel_row_len()
    _, e0 = eng.panel_scale()
    eng.panel_scale(set_e0max=float(comm.allreduce_max(e0)[0]))
    hist = torch.zeros(eng.NBINS, dtype=torch.int32, device=dev)
    rows_loc = torch.zeros(cap * rl, dtype=torch.float64, device=dev)
    idx_loc = torch.full((cap,), -1, dtype=torch.int64, device=dev)
    global_all = n_rows <= cap
    k, stalled, forced_next = 0, 0, -1
    while k < K:
        forced = forced_next if stalled >= 2 else -1
        if forced < 0 and not global_all:
            for level in (1, 2):
                eng.panel_hist(level, hist.data_ptr())
                comm.allreduce_tensor(hist)
                eng.panel_tau(level, hist.data_ptr())
        take_all = forced >= 0 or global_all
        n_c, ov = eng.panel_select(k, rows_loc.data_ptr(), idx_loc.data_ptr(), forced, take_all)
        info = comm.all_gather_ints([n_c, int(ov)])
        counts, total = info[:, 0].copy(), int(info[:, 0].sum())
        done = 0
        if not info[:, 1].any() and 0 < total <= cap:
            maxc = int(counts.max())
            rows_g = torch.empty(comm.world * maxc * rl, dtype=torch.float64, device=dev)
            idx_g = torch.empty(comm.world * maxc, dtype=torch.int64, device=dev)
            comm.all_gather_into(rows_g, rows_loc[:maxc * rl])
            comm.all_gather_into(idx_g, idx_loc[:maxc])
            eng.panel_assemble(rows_g.data_ptr(), idx_g.data_ptr(), counts, maxc)
            steps = 1 if forced >= 0 else min(16, K - k)
            done = eng.panel_run(k, steps, take_all)
        if done == 0:
            # nothing provable (stale bound / exact ties): exact energies everywhere, retry; a second
            # failure forces the global first arg-max as the only candidate
            stalled += 1
            if stalled > 3:
                raise ArithmeticError("deflation made no progress at component %d" % k)
            e, g = eng.panel_refresh(k)
            both = comm.allreduce_max(np.eye(comm.world)[comm.rank] * e) if comm.multi else np.array([e])
            gids = comm.all_gather_ints([g])[:, 0]
            order = sorted(range(comm.world), key=lambda r: (-both[r], gids[r]))
            forced_next = int(gids[order[0]])
            continue
        stalled = 0
        eng.panel_project(k, done)
        k += done

