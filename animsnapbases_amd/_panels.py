"""Projection-mode ("panel") deflation over several ranks -- shared by posComponents (vertex rows) and
constraintsComponents 'pca_blocks' with p = 1 (constraint rows)."""
import numpy as np


def _residual_rest(eng, comm, k0, K):
    """Components k0 .. K - 1 through the residual protocol (posComponents.py:76-96 as is: one all-gather of
    [energy, idx, 3 x F slab] records per component) -- where a run ends up that left the projection mode."""
    rec = recs = None
    if comm.multi:
        rec, recs = comm.new_records(eng.xchg_len(), eng.device_exchange)
    for k in range(k0, K):
        if comm.multi:
            eng.local_best(k, rec.data_ptr())
            comm.all_gather_records(rec, recs)
            eng.pick(k, recs.data_ptr(), comm.world)
        else:
            eng.pick(k)
        eng.apply(k)


def deflate_panels_multirank(eng, comm, n_rows, K):
    """Projection-mode deflation over several ranks.  Per PANEL (up to 16 components):
      1. every rank thresholds its OWN energies (two local histogram steps, no collective) and exports its ~768 largest
         energies + its local threshold; ONE small all-gather (12 KB per rank) gives every rank the same global
         threshold tau -- the (m_target+1)-th largest energy overall, never below a rank's local bound -- and the
         per-rank candidate counts;
      2. each rank rebuilds the exact residual rows of its own candidates (energy > tau); a padded all-gather of
         rows and of vertex ids replicates the ~768 candidate rows on every rank;
      3. every rank runs the identical greedy steps on them (no per-component collective), then projects its shard.
    Two collectives (energies; rows with their vertex ids packed behind them) and two host synchronisations per panel.
    Ranks stay in lock-step because every decision is a function of all-gathered data.
    Unproven steps (ASB_SPEC_PANELS, default on): when the bound on the vertices outside the candidate set is too stale
    to prove a winner, the panel goes on with the exact winner among the candidates (identical on every rank); the
    pass over X then checks those steps against every vertex's energy, each rank on its shard, and one extra tiny
    all-reduce (min) fixes how many of them stand -- fewer, longer panels for one more collective on such panels."""
    import os
    dev = comm.exchange_device(eng.device_exchange)
    torch = comm._torch
    cap, rl = eng.panel_capacity(), eng.panel_row_len()
    m_target = eng.panel_target()
    _, e0 = eng.panel_scale()
    # ONE start-up exchange: the largest initial energy (histogram range and rounding margin must be the same everywhere) and
    # what decides the guessed first panel travel together, as bit patterns in one all-gather of four words per rank
    gs = (0.0, 0.0, True)
    want_guess = hasattr(eng, "panel_guess_stats") and os.environ.get("ASB_FIRST_PANEL_MEAN", "1") != "0"
    if want_guess:
        gs = eng.panel_guess_stats()
    start = np.array([e0, gs[0], gs[1], 0.0 if gs[2] else 1.0], dtype=np.float64)
    allst = comm.all_gather_ints(start.view(np.int64)).view(np.float64).reshape(-1, 4)
    eng.panel_scale(set_e0max=float(allst[:, 0].max()))
    # exchange buffers, kept on the engine between calls (49 MB: allocating and clearing them cost 1.3 ms per call).
    # Nothing reads the padding: the assembly takes counts[r] rows of rank r's piece.
    key = (cap, rl, comm.world, str(dev))
    bufs = getattr(eng, "_panel_bufs", None)
    if bufs is None or bufs[0] != key:
        bufs = (key, torch.empty(cap * (rl + 1), dtype=torch.float64, device=dev),      # rows, then (packed exchange) the ids
                torch.empty(cap, dtype=torch.int64, device=dev),
                torch.empty(cap + 1, dtype=torch.float64, device=dev),
                torch.empty(comm.world * (cap + 1), dtype=torch.float64, device=dev))
        try:
            eng._panel_bufs = bufs
        except AttributeError:
            pass
    _, rows_loc, idx_loc, top_loc, top_all = bufs
    packed_ok = hasattr(eng, "panel_assemble_packed")
    spec_word = torch.zeros(1, dtype=torch.float64, device=dev) if (hasattr(eng, "panel_project_spec_dev") and dev.type != "cpu") else None
    # the reduced word comes back through the engine's polled pinned slot when the collective ran on the engine's stream
    # (a communicator that cannot say so -- or an engine on another stream -- gets the synchronising read)
    oes = getattr(comm, "on_engine_stream", None)
    same_stream = spec_word is not None and hasattr(eng, "fetch_double") and bool(oes(eng) if callable(oes) else oes)
    read_word = (lambda: eng.fetch_double(spec_word.data_ptr())) if same_stream else (lambda: spec_word.item())
    # lock-step protection of the co-resident panel kernel (see below): only with the real engine, several ranks, kernel on
    coop_check = bool(spec_word is not None and comm.multi and hasattr(eng, "panel_set_coop") and
                      os.environ.get("ASB_PANEL_COOP", "1") != "0")
    global_all = n_rows <= cap
    spec_budget = 16 if (hasattr(eng, "panel_run_spec") and os.environ.get("ASB_SPEC_PANELS", "1") != "0") else 0
    # first panel guessed from the energies without the constant-in-time direction (asb.h: asb_panel_guess_*): a collective
    # decision -- every rank must be able to, and the share of that direction in |X|^2 over ALL shards must exceed 1/4
    guess_ok = False
    if spec_budget and not global_all and want_guess:
        tot = allst[:, 1:].sum(axis=0)
        guess_ok = bool(tot[2] == 0 and tot[1] > 0 and tot[0] > 0.25 * tot[1])
    guessing = False
    # several sub-panels per read (see the loop): only with the real engine, the co-resident kernel and device-side counts
    multi_sub = bool(spec_word is not None and hasattr(eng, "panel_sub_run") and hasattr(eng, "panel_set_coop") and
                     os.environ.get("ASB_PANEL_COOP", "1") != "0" and os.environ.get("ASB_DOUBLE_PANELS", "1") != "0")
    sub_max = max(1, min(4, int(os.environ.get("ASB_SUB_PANELS", "4"))))
    sub_cur = min(sub_max, max(1, int(os.environ.get("ASB_SUB_FIRST", "4"))))
    sub_budget = [16] * 8
    # the read in one launch + one exchange (round 4); ASB_SUB_CHAIN=0: one launch and one exchange per sub-panel / tile (round 2)
    one_launch = bool(multi_sub and hasattr(eng, "panel_read_run") and os.environ.get("ASB_SUB_CHAIN", "1") != "0")
    words = torch.zeros(10, dtype=torch.float64, device=dev) if one_launch else None
    n_collectives = [0]
    # the stall rule of the single-rank driver (asb.h: asb_project_switch_residual): reads of X that commit fewer than 3/4 of a
    # component each (K beyond the numerical rank: every panel ends in a refresh) -- the run continues in the residual protocol
    stall_rule = hasattr(eng, "project_switch_residual") and os.environ.get("ASB_STALL_FALLBACK", "1") != "0"
    mark_reads = mark_k = reads = 0
    k, stalled, forced_next = 0, 0, -1
    while k < K:
        if guessing:
            eng.panel_guess_end()
            guessing = False
        if stall_rule and reads - mark_reads >= 8:           # (every quantity below is the same on all ranks)
            slow = (k - mark_k) * 4 < (reads - mark_reads) * 3
            mark_reads, mark_k = reads, k
            if slow:
                eng.project_switch_residual(k)
                _residual_rest(eng, comm, k, K)
                return k
        reads += 1
        forced = forced_next if stalled >= 2 else -1
        take_all = forced >= 0 or global_all
        if not take_all:
            if guess_ok and k == 0 and stalled == 0:
                eng.panel_guess_begin(comm.world)
                guessing = True
            for level in (1, 2):                         # local threshold: ~m_target of this rank's vertices above it
                eng.panel_hist(level, None)
                eng.panel_tau(level, None)
            eng.panel_top_energies(top_loc.data_ptr(), cap)
            comm.all_gather_into(top_all, top_loc)
            counts = eng.panel_global_tau(top_all.data_ptr(), comm.world, cap)      # tau installed; the panel's first host sync
            if counts is None:                            # table too large for the selection kernel: same rule with torch
                tab = top_all.view(comm.world, cap + 1)
                exported = tab[:, :cap].reshape(-1)
                kth = torch.topk(exported, eng.panel_target() + 1).values[-1].clamp(min=0.0)
                tau = torch.maximum(kth, tab[:, cap].max()).reshape(1).contiguous()
                eng.panel_set_tau(tau.data_ptr())
                counts = (tab[:, :cap] > tau).sum(dim=1).cpu().numpy().astype(np.int64)
            packed = packed_ok and not guessing and 0 < int(counts.sum()) <= cap
            if guessing:    # the union's size is not in the gathered energies: one more small exchange, first panel only
                n_c, ov = eng.panel_select(k, rows_loc.data_ptr(), idx_loc.data_ptr(), -1, False)
                info = comm.all_gather_ints([n_c, int(ov)])
                counts = info[:, 0].copy()
            elif packed:      # the ids go right behind this rank's maxc rows: rows and ids travel in ONE all-gather
                maxc = int(counts.max())
                eng.panel_select(k, rows_loc.data_ptr(), rows_loc.data_ptr() + 8 * maxc * rl, -1, False, want_counts=False)
            else:
                eng.panel_select(k, rows_loc.data_ptr(), idx_loc.data_ptr(), -1, False, want_counts=False)
            overflow = bool(guessing and info[:, 1].any())
        else:
            packed = False
            n_c, ov = eng.panel_select(k, rows_loc.data_ptr(), idx_loc.data_ptr(), forced, True)
            info = comm.all_gather_ints([n_c, int(ov)])
            counts, overflow = info[:, 0].copy(), bool(info[:, 1].any())
        total = int(counts.sum())
        done = 0
        if not overflow and 0 < total <= cap:
            maxc = int(counts.max())
            if packed:
                piece = maxc * (rl + 1)
                buf_g = torch.empty(comm.world * piece, dtype=torch.float64, device=dev)
                comm.all_gather_into(buf_g, rows_loc[:piece])
                eng.panel_assemble_packed(buf_g.data_ptr(), counts, maxc)
            else:
                rows_g = torch.empty(comm.world * maxc * rl, dtype=torch.float64, device=dev)
                idx_g = torch.empty(comm.world * maxc, dtype=torch.int64, device=dev)
                comm.all_gather_into(rows_g, rows_loc[:maxc * rl])
                comm.all_gather_into(idx_g, idx_loc[:maxc])
                eng.panel_assemble(rows_g.data_ptr(), idx_g.data_ptr(), counts, maxc)
            steps = 1 if forced >= 0 else min(16, K - k)
            # Several sub-panels per read of X (what the single-rank driver does inside the library): up to three runs of
            # the panel kernel on the same candidates, ONE pass over the shard for all their columns, the tiles checked one
            # at a time with a min over the ranks in between.  Every rank sees the same candidates, so the runs are
            # identical everywhere -- except that the kernel can time out on one rank: the first exchange carries the
            # status, and on a failure all ranks switch the kernel off and repeat the panel the plain way.
            handled = multi_sub and spec_budget and not take_all and stalled == 0 and K - k > 16
            if handled and one_launch:
                # ONE launch of the panel kernel for all sub-panels of the read (every rank holds the same candidates, so the
                # runs and their nine-word summary are identical everywhere), this shard's pass and the checks of all its tiles
                # enqueued behind it, ONE min-all-reduce of the per-tile counts (+ the kernel's status) for the whole read, one
                # host read (asb.h: asb_panel_read_run / _commit).  Per read of X: two all-gathers (energies; rows + ids) and
                # this all-reduce.
                n_collectives[0] += 1
                nt, ncs, provs = eng.panel_read_run(k, K, sub_cur, spec_budget, sub_budget, words.data_ptr())
                comm.allreduce_min_tensor(words[:9])
                w10 = eng.fetch_doubles(words.data_ptr(), 10) if same_stream else words.cpu().numpy()
                if w10[8] < 0:                                # somewhere the exchange timed out: all ranks leave the kernel
                    w10[:9] = 0.0
                    eng.panel_read_commit(w10)                # (rolls back what this shard's local chain adopted; commits nothing)
                    eng.panel_set_coop(False)
                    coop_check = multi_sub = guess_ok = False     # (unproven steps, and with them the guess, need the kernel)
                    continue                                  # the panel again, from the selection, the plain way
                total, full, rejected = eng.panel_read_commit(w10)
                if nt > 0:
                    for ct in range(1, nt):
                        if ct <= full:
                            sub_budget[ct] = min(16, max(4, int(w10[ct]) + 2))
                    if rejected:
                        sub_cur = min(sub_max, full + 1)
                    elif nt == sub_cur:
                        sub_cur = min(sub_max, 2 * sub_cur)
                    gain = total - provs[0]
                    spec_budget = 16 if gain > 0 else max(2, spec_budget // 2)
                if total > 0:
                    stalled = 0
                    k += total
                    continue
                done = 0                                      # nothing stood: the refresh below
            if handled and not one_launch:
                tiles, failed = [], False
                for sp in range(sub_cur):
                    kb = k + 16 * sp
                    if kb >= K:
                        break
                    st_ = min(16, K - kb) if sp == 0 else min(16, K - kb, sub_budget[sp])
                    ran, proven, cont = eng.panel_sub_run(sp, kb, st_, spec_budget if sp == 0 else 16)
                    if ran < 0:
                        failed = True
                        break
                    if ran == 0:
                        break
                    tiles.append((kb, ran, proven))
                    if not cont:
                        break
                spec_word.fill_(-1.0 if failed else float(len(tiles)))
                comm.allreduce_min_tensor(spec_word)
                agreed = int(read_word())
                if agreed < 0:                                # somewhere the exchange timed out: all ranks leave the kernel
                    eng.panel_set_coop(False)
                    coop_check = multi_sub = guess_ok = False     # (unproven steps, and with them the guess, need the kernel)
                    continue                                  # the panel again, from the selection, the plain way
                total = 0
                tiles = tiles[:agreed]                        # (identical on every rank by construction; the agreed count keeps the
                if agreed > 0:                                # number of collectives below identical in any case)
                    eng.panel_sub_project(k, [t[1] for t in tiles])
                    full, rejected = 0, False
                    for ct, (kb, ran, proven) in enumerate(tiles):
                        eng.panel_sub_check(ct, kb, ran, spec_word.data_ptr())
                        comm.allreduce_min_tensor(spec_word)
                        kept = int(read_word())
                        eng.panel_sub_commit(ct, kb, ran, kept)
                        total += kept
                        if ct >= 1:
                            sub_budget[ct] = min(16, max(4, kept + 2))
                        if kept < ran:
                            rejected = True
                            break
                        full += 1
                    if rejected:
                        sub_cur = min(sub_max, full + 1)
                    elif len(tiles) == sub_cur:
                        sub_cur = min(sub_max, 2 * sub_cur)
                    gain = total - tiles[0][2]
                    spec_budget = 16 if gain > 0 else max(2, spec_budget // 2)
                if total > 0:
                    stalled = 0
                    k += total
                    continue
                done = 0                                      # nothing stood: the refresh below
            if not handled:
                while True:
                    if spec_budget and not take_all and stalled == 0:
                        done, proven = eng.panel_run_spec(k, steps, take_all, spec_budget)
                    else:
                        done = proven = eng.panel_run(k, steps, take_all)
                    if not coop_check:
                        if done < 0:          # one rank (tests): the context has switched the timed-out kernel off; repeat
                            continue
                        break
                    # The co-resident panel kernel can time out on ONE rank (its GPU shared with other work).  Every rank must
                    # then redo the panel the same way -- the two-kernel loop, whose steps are the provable ones -- or the
                    # ranks fall out of lock-step.  The status rides on the min all-reduce that panels with unproven steps
                    # need anyway: a rank whose launch failed contributes -1 (and skips its pass).
                    if done > proven and done > 0:
                        eng.panel_project_spec_dev(k, done, proven, spec_word.data_ptr())
                        passed = True
                    else:
                        spec_word.fill_(float(done))             # -1: failed here; otherwise the fully proven count
                        passed = False
                    comm.allreduce_min_tensor(spec_word)
                    agreed = int(read_word())
                    if agreed >= 0:
                        break
                    eng.panel_set_coop(False)                    # somewhere the exchange timed out: all ranks leave the kernel
                    coop_check = multi_sub = False
                if coop_check:
                    if passed:                                    # unproven tail: `agreed` of the steps stand on every shard
                        done = agreed
                        eng.panel_commit(k, done)
                        gain = done - proven
                        spec_budget = 16 if gain > 0 else max(2, spec_budget // 2)
                        if done > 0:
                            stalled = 0
                            k += done
                            continue
                    # fully proven (or nothing ran): falls through to the plain pass / the refresh below
                elif done > proven:                           # the tail is unproven: the pass decides how much of it stands
                    if spec_word is not None:                 # count stays on the device: min over ranks, ONE read
                        eng.panel_project_spec_dev(k, done, proven, spec_word.data_ptr())
                        comm.allreduce_min_tensor(spec_word)
                        done = int(read_word())
                    else:
                        mine = eng.panel_project_spec(k, done, proven)
                        done = int(-comm.allreduce_max(np.array([-float(mine)]))[0]) if comm.multi else mine
                    eng.panel_commit(k, done)
                    gain = done - proven
                    # a kept step saves 1/16 of a panel, a rejected one costs one step of the panel kernel: back off only
                    # after complete failures
                    spec_budget = 16 if gain > 0 else max(2, spec_budget // 2)
                    if done > 0:
                        stalled = 0
                        k += done
                        continue
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
    if guessing:
        eng.panel_guess_end()
    return K
