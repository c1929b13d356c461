"""Launch schedules of the training scan, checked WITHOUT a GPU: parrot_decoder_trace makes a plan record, for every
launch and every job in it, the byte ranges it reads and writes (on fake device addresses; nothing is launched).  The
orderings a schedule has to keep follow from the scan of model.py:651-737 and its gradient:

  * no job of a launch touches a range another job of the same launch writes (the jobs of a launch run concurrently) --
    except a read the job takes behind the in-launch flag of schedule 7, whose writer must then be the attention job;
  * a write-once buffer (states, gates, window parameters, saved activations, pre-activation scratch; in the backward the
    pre-activation gradients) is written exactly once per element and read only by LATER launches;
  * an accumulator (dh, dw, dw0, dhup) is never written again after a job has taken it as a plain input (the state /
    attention backward consuming the total).

Run for schedules 0, 5 and 7, GRU and LSTM layers, 1-3 layers, with and without caller data in the per-step input
buffers, forward and backward."""
import ctypes as C
import os

import pytest


def _lib():
    from parrot_amd import _lib as L
    try:
        return L, L.load()
    except L.HipLibraryMissing:
        pytest.skip("libparrot_hip.so not built")


class _Arena:
    """Fake device address space: distinct, 4 KB aligned ranges; name -> (lo, hi)."""

    def __init__(self):
        self.top = 0x10000000
        self.ranges = {}

    def take(self, name, nbytes):
        lo = self.top
        self.top = (lo + max(nbytes, 16) + 4095) // 4096 * 4096 + 4096
        self.ranges[name] = (lo, lo + nbytes)
        return lo


def _make_plan(L, lib, sched, cell, nl, seq_init, monkeypatch, T=5, B=20, H=32, E=16, A=4, U=7, bf16=0, hetero=False):
    if sched is None:
        monkeypatch.delenv("PARROT_SCHEDULE", raising=False)
    else:
        monkeypatch.setenv("PARROT_SCHEDULE", str(sched))
    monkeypatch.setenv("PARROT_TRACE_ONLY", "1")
    ar = _Arena()
    d = L.DecoderDesc()
    d.T, d.B, d.H, d.E, d.A, d.U, d.L = T, B, H, E, A, U, nl
    d.cell, d.use_graph, d.seq_init, d.bf16 = cell, 0, seq_init, bf16
    d.eps, d.alignment, d.sharpening, d.timing = 1e-5, 1.0, 1.0, 1.0
    f = 4
    gw = 4 * H if cell == 1 else 2 * H
    for l in range(nl):
        K = H + E + l * H
        d.Wg[l] = ar.take(f"Wg{l}", K * gw * f)
        d.bg[l] = ar.take(f"bg{l}", gw * f)
        d.Wg_f[l] = ar.take(f"Wg_f{l}", K * gw * f)
        d.Wg_r[l] = ar.take(f"Wg_r{l}", K * gw * f)
        d.h[l] = ar.take(f"h{l}", (T + 1) * B * H * f)
        d.dh[l] = ar.take(f"dh{l}", (T + 1) * B * H * f)
        d.dG[l] = ar.take(f"dG{l}", T * B * gw * f)
        if l < nl - 1:
            d.dhup[l] = ar.take(f"dhup{l}", (T + 1) * B * H * f)
        if bf16 and cell == 1:  # bf16 copies of the pre-activation gradients, written by the fused backward tick (round 5)
            d.dG16[l] = ar.take(f"dG16_{l}", T * B * gw * 2)
        if (bf16 and cell == 1) or hetero:  # second (and third) accumulators: the backward products run as K parts
            d.dh_b[l] = ar.take(f"dh_b{l}", (T + 1) * B * H * f)
            if l < nl - 1:
                d.dhup_b[l] = ar.take(f"dhup_b{l}", (T + 1) * B * H * f)
                if hetero:
                    d.dhup_c[l] = ar.take(f"dhup_c{l}", (T + 1) * B * H * f)
        if l >= 1 or (seq_init >> l) & 1:
            d.seq_g[l] = ar.take(f"seq_g{l}", T * B * gw * f)
        if cell == 0:
            d.Wc[l] = ar.take(f"Wc{l}", K * H * f)
            d.bc[l] = ar.take(f"bc{l}", H * f)
            d.Wc_f[l] = ar.take(f"Wc_f{l}", K * H * f)
            d.Wc_r[l] = ar.take(f"Wc_r{l}", K * H * f)
            for n in ("z", "r", "rh", "c"):
                getattr(d, n)[l] = ar.take(f"{n}{l}", T * B * H * f)
            d.dC[l] = ar.take(f"dC{l}", T * B * H * f)
            if l >= 1 or (seq_init >> l) & 1:
                d.seq_c[l] = ar.take(f"seq_c{l}", T * B * H * f)
        else:
            d.cst[l] = ar.take(f"cst{l}", (T + 1) * B * H * f)
            d.gate4[l] = ar.take(f"gate4{l}", T * B * 4 * H * f)
            d.dcell[l] = ar.take(f"dcell{l}", B * H * f)
    d.WattT = ar.take("WattT", 3 * A * H * f)
    d.batt = ar.take("batt", 3 * A * f)
    d.ctx = ar.take("ctx", B * U * E * f)
    d.w = ar.take("w", (T + 1) * B * E * f)
    d.kappa = ar.take("kappa", (T + 1) * B * A * f)
    d.a = ar.take("a", T * B * A * f)
    d.b = ar.take("b", T * B * A * f)
    d.phi = ar.take("phi", T * B * U * f)
    d.dw = ar.take("dw", (T + 1) * B * E * f)
    d.dw0 = ar.take("dw0", (T + 1) * B * E * f)
    if (bf16 and cell == 1) or hetero:
        d.dw_b = ar.take("dw_b", (T + 1) * B * E * f)
        d.dw0_b = ar.take("dw0_b", (T + 1) * B * E * f)
        if hetero:
            d.dw_c = ar.take("dw_c", (T + 1) * B * E * f)
            d.dw0_c = ar.take("dw0_c", (T + 1) * B * E * f)
    d.dkappa = ar.take("dkappa", B * A * f)
    d.dp = ar.take("dp", T * B * 3 * A * f)
    d.att_sup = ar.take("att_sup", T * B * 2 * 4)
    plan = C.c_void_p()
    rc = lib.parrot_decoder_create(C.byref(d), C.byref(plan))
    assert rc == 0, rc
    return plan, ar, d


def _trace(lib, plan, which):
    n = lib.parrot_decoder_trace(plan, which, None, 0)
    assert n > 0, n
    buf = (C.c_longlong * (5 * n))()
    assert lib.parrot_decoder_trace(plan, which, buf, n) == n
    recs = [tuple(buf[5 * i:5 * i + 5]) for i in range(n)]  # (launch, job, kind, lo, hi)
    return recs


def _owner(ar, lo):
    for name, (a, b) in ar.ranges.items():
        if a <= lo < b:
            return name
    return None


def _overlap(a, b):
    return a[0] < b[1] and b[0] < a[1]


def _check(recs, ar, write_once, accumulators, T, slot_bytes):
    by_launch = {}
    for launch, job, kind, lo, hi in recs:
        assert hi > lo
        assert _owner(ar, lo) is not None and _owner(ar, lo) == _owner(ar, hi - 1), "range outside any buffer"
        by_launch.setdefault(launch, []).append((job, kind, lo, hi))
    # ---- 1. jobs of one launch do not interfere
    for launch, rs in by_launch.items():
        writes = [(job, lo, hi) for job, kind, lo, hi in rs if kind in (1, 2)]
        for job, kind, lo, hi in rs:
            for wjob, wlo, whi in writes:
                if wjob == job or not _overlap((lo, hi), (wlo, whi)):
                    continue
                if kind == 3:  # producers of an in-launch hand-off: the attention job (100) or a state-backward chain (200+)
                    assert wjob >= 100, f"launch {launch}: flagged read of {_owner(ar, lo)} written by job {wjob}"
                    continue
                raise AssertionError(f"launch {launch}: job {job} ({'rwx?'[kind]}) and job {wjob} (w) overlap in "
                                     f"{_owner(ar, lo)} [{lo:#x}, {hi:#x})")
        flagged = [(lo, hi) for job, kind, lo, hi in rs if kind == 3]
        for lo, hi in flagged:  # a flag without its producer in the launch would wait for ever
            assert any(wjob >= 100 and _overlap((lo, hi), (wlo, whi)) for wjob, wlo, whi in writes), launch
    # ---- 2. write-once buffers: one writer per byte, readers strictly later (or behind the flag in the same launch)
    first_write, last_write = {}, {}  # (name, granule) -> launch
    G = 4  # bytes (one float)
    def granules(lo, hi):
        return range(lo // G, (hi + G - 1) // G)
    for launch in sorted(by_launch):
        for job, kind, lo, hi in by_launch[launch]:
            name = _owner(ar, lo)
            if name not in write_once or kind not in (1, 2):
                continue
            assert kind == 1 or name.startswith("seq_"), f"{name} is write-once but launch {launch} updates it in place"
            for g in granules(lo, hi):
                if kind == 1:
                    assert (name, g) not in first_write, f"{name} written twice (launches {first_write.get((name, g))}, {launch})"
                first_write.setdefault((name, g), launch)
                last_write[(name, g)] = launch
    for launch in sorted(by_launch):
        for job, kind, lo, hi in by_launch[launch]:
            name = _owner(ar, lo)
            if name not in write_once or kind not in (0, 3):
                continue
            for g in granules(lo, hi):
                w = first_write.get((name, g))
                if w is None:
                    # slot 0 of the histories = the state entering the window (caller's); everything else must be written
                    a0 = ar.ranges[name][0]
                    assert name in slot_bytes and g * G < a0 + slot_bytes[name], f"{name} read at launch {launch} but never written"
                    continue
                if kind == 3:
                    assert w == launch
                else:  # (after ALL writes: the two-part input projections of LSTM stacks update their scratch in place)
                    assert last_write[(name, g)] < launch, f"{name}: read in launch {launch}, written in launch {last_write[(name, g)]}"
    # ---- 3. accumulators: no write after a plain read
    last_plain_read = {}
    for launch in sorted(by_launch):
        for job, kind, lo, hi in by_launch[launch]:
            name = _owner(ar, lo)
            if name not in accumulators:
                continue
            for g in granules(lo, hi):
                if kind in (1, 2):
                    r = last_plain_read.get((name, g))
                    assert r is None or r[0] > launch or (r[0] == launch and r[1] == job), \
                        f"{name}: written in launch {launch} after launch {r[0]} had consumed it"
                if kind == 0:
                    last_plain_read[(name, g)] = (launch, job)
    return len(by_launch)


@pytest.mark.parametrize("sched", [0, 5, 7])
@pytest.mark.parametrize("cell,nl,seq_init", [(0, 1, 0), (0, 2, 0), (0, 3, 0), (0, 3, 0b101), (1, 1, 0), (1, 2, 0), (1, 3, 0b010)])
def test_schedule_keeps_the_scan_orderings(monkeypatch, sched, cell, nl, seq_init):
    L, lib = _lib()
    T, B, H, E, A, U = 5, 20, 32, 16, 4, 7
    plan, ar, d = _make_plan(L, lib, sched, cell, nl, seq_init, monkeypatch, T, B, H, E, A, U)
    try:
        got = lib.parrot_decoder_schedule(plan)
        want = sched
        if sched == 7 and cell == 0:
            want = 5        # one launch per tick with the attention inside it: LSTM layers only
        if want == 5 and nl == 1:
            want = 0
        assert got == want, (got, want)
        f = 4
        slot = {"w": B * E * f, "kappa": B * A * f}
        fwd_once = {"w", "kappa", "a", "b", "phi", "att_sup"}
        for l in range(nl):
            slot[f"h{l}"] = B * H * f
            fwd_once |= {f"h{l}"}
            if cell == 0:
                fwd_once |= {f"z{l}", f"r{l}", f"rh{l}", f"c{l}"}
            else:
                slot[f"cst{l}"] = B * H * f
                fwd_once |= {f"cst{l}", f"gate4{l}"}
            if l >= 1 and got >= 5:  # the input projections' scratch (caller data is accumulated onto, never rewritten)
                fwd_once |= {f"seq_g{l}"} | ({f"seq_c{l}"} if cell == 0 else set())
        recs = _trace(lib, plan, 0)
        if got >= 5:  # buffers with caller data are read-modify-write targets of the input projections: whole buffer = "slot"
            for l in range(1, nl):
                if (seq_init >> l) & 1:
                    gw = 4 * H if cell == 1 else 2 * H
                    slot[f"seq_g{l}"] = T * B * gw * f
                    if cell == 0:
                        slot[f"seq_c{l}"] = T * B * H * f
        n_fwd = _check(recs, ar, fwd_once, set(), T, slot)
        ticks = {0: T + nl - 1, 5: T + nl, 7: T + max(1, nl if nl > 1 else 0)}[got]
        per_tick = 1 if got == 7 else (2 if cell == 1 else 3)
        assert n_fwd <= ticks * per_tick and n_fwd >= T * per_tick - 2, (n_fwd, ticks, per_tick)
        # backward
        bwd_once = {"dp"} | {f"dG{l}" for l in range(nl)} | ({f"dC{l}" for l in range(nl)} if cell == 0 else set())
        acc = {"dw", "dw0"} | {f"dh{l}" for l in range(nl)} | {f"dhup{l}" for l in range(nl - 1)}
        recs = _trace(lib, plan, 1)
        _check(recs, ar, bwd_once, acc, T, {})
    finally:
        lib.parrot_decoder_destroy(plan)


@pytest.mark.parametrize("cell,nl", [(0, 2), (0, 3), (1, 2), (1, 3)])
def test_schedule5_w_rows_ride_in_the_step_jobs(monkeypatch, cell, nl):
    """Round 6 (plans.hip fwd5, s5_w_in_step): the rows of w of an upper layer's input projection (K = E) are a second segment
    of that layer's own gate / candidate job; the attention launch's projection jobs walk the h rows only.  Both placements
    (PARROT_S5_WSTEP=0: the round-5 one) keep every read-after-write ordering; with the new one no projection job walks
    K = E + l H any more and the upper layers' step jobs walk K = H + E."""
    L, lib = _lib()
    T, B, H, E, A, U = 6, 20, 32, 16, 4, 7
    ks = {}
    for wstep in ("1", "0"):
        monkeypatch.setenv("PARROT_S5_WSTEP", wstep)
        plan, ar, d = _make_plan(L, lib, 5, cell, nl, 0, monkeypatch, T, B, H, E, A, U)
        try:
            assert lib.parrot_decoder_schedule(plan) == 5
            f = 4
            slot = {"w": B * E * f, "kappa": B * A * f}
            fwd_once = {"w", "kappa", "a", "b", "phi", "att_sup"}
            for l in range(nl):
                slot[f"h{l}"] = B * H * f
                fwd_once |= {f"h{l}"}
                if cell == 0:
                    fwd_once |= {f"z{l}", f"r{l}", f"rh{l}", f"c{l}"}
                else:
                    slot[f"cst{l}"] = B * H * f
                    fwd_once |= {f"cst{l}", f"gate4{l}"}
                if l >= 1:
                    fwd_once |= {f"seq_g{l}"} | ({f"seq_c{l}"} if cell == 0 else set())
            _check(_trace(lib, plan, 0), ar, fwd_once, set(), T, slot)
            n = lib.parrot_decoder_trace_jobs(plan, 0, None, 0)
            buf = (C.c_longlong * (6 * n))()
            lib.parrot_decoder_trace_jobs(plan, 0, buf, n)
            jobs = [tuple(buf[6 * i:6 * i + 6]) for i in range(n)]  # (launch, job, M, N, Ksum, epi)
            ks[wstep] = (sorted({j[4] for j in jobs if j[5] == 0}), sorted({j[4] for j in jobs if j[5] > 0}))
        finally:
            lib.parrot_decoder_destroy(plan)
    lin_new, step_new = ks["1"]
    lin_old, step_old = ks["0"]
    assert max(lin_old) == E + (nl - 1) * H or (nl == 3 and max(lin_old) >= E + H)  # the round-5 projections walk the w rows
    assert all(k % H == 0 for k in lin_new), lin_new  # ... the new ones the h rows only
    assert H + E in step_new and H in step_old and H not in step_new, (step_new, step_old)


@pytest.mark.parametrize("nl", [1, 2, 3])
def test_fused_lstm_ticks_keep_the_scan_orderings(monkeypatch, nl):
    """bf16 LSTM stacks the wide kernel takes run schedule 7 by default, with BOTH ticks as one launch: forward, the
    attention at the head of the next tick's launch (layer 0's w rows behind the flag); backward, the attention backward
    and the state-backward rows at the head of the launch whose products read the dP rows they publish (every such read
    is behind a flag whose writer -- the attention job or a chain -- is in the same launch)."""
    L, lib = _lib()
    monkeypatch.setenv("PARROT_WK", "2")
    T, B, H, E, A, U = 5, 20, 64, 64, 4, 7
    plan, ar, d = _make_plan(L, lib, None, 1, nl, 0, monkeypatch, T, B, H, E, A, U, bf16=1)
    try:
        assert lib.parrot_decoder_schedule(plan) == 7
        f = 4
        slot = {"w": B * E * f, "kappa": B * A * f}
        fwd_once = {"w", "kappa", "a", "b", "phi", "att_sup"}
        for l in range(nl):
            slot[f"h{l}"] = B * H * f
            slot[f"cst{l}"] = B * H * f
            fwd_once |= {f"h{l}", f"cst{l}", f"gate4{l}"}
        n_fwd = _check(_trace(lib, plan, 0), ar, fwd_once, set(), T, slot)
        assert n_fwd == T + max(1, nl if nl > 1 else 0)
        recs = _trace(lib, plan, 1)
        assert lib.parrot_decoder_writes_bf16_grads(plan) == 1
        bwd_once = {"dp"} | {f"dG{l}" for l in range(nl)} | {f"dG16_{l}" for l in range(nl)}
        acc = {"dw", "dw0", "dw_b", "dw0_b"} | {f"dh{l}" for l in range(nl)} | {f"dhup{l}" for l in range(nl - 1)}
        acc |= {f"dh_b{l}" for l in range(nl)} | {f"dhup_b{l}" for l in range(nl - 1)}
        n_bwd = _check(recs, ar, bwd_once, acc, T, {})
        assert n_bwd == T + nl - 1  # one launch per tick
        flagged = [r for r in recs if r[2] == 3]
        assert flagged and all(_owner(ar, r[3]).startswith("dG") for r in flagged)
        # the products really run as two K halves: the second accumulators are written (stored) and read back
        for l in range(nl):  # every row of the bf16 gradient copies is written (once), nothing in the scan reads them
            kinds = {r[2] for r in recs if _owner(ar, r[3]) == f"dG16_{l}"}
            assert kinds == {1}, (l, kinds)
        for name in ["dw0_b"] + [f"dh_b{l}" for l in range(nl)] + [f"dhup_b{l}" for l in range(nl - 1)]:
            kinds = {r[2] for r in recs if _owner(ar, r[3]) == name}
            assert (1 in kinds or 2 in kinds) and 0 in kinds, (name, kinds)
    finally:
        lib.parrot_decoder_destroy(plan)


@pytest.mark.parametrize("nl", [2, 3])
@pytest.mark.parametrize("sched", [5, 0])
def test_k_balanced_backward_tick_keeps_the_scan_orderings(monkeypatch, sched, nl):
    """bwd8 (2-layer f32 GRU decoders with all accumulators given): every K = 2H product of a backward tick in two K = H
    halves with their own buffers, layer 1 two ticks ahead of layer 0, its dG-fed downward products in the NEXT tick's
    heterogeneous attention launch.  Same ordering rules as every other schedule; plus: three launches per tick, no job
    walks more than K = H, and the attention launch carries step-GEMM jobs."""
    L, lib = _lib()
    T, B, H, E, A, U = 6, 20, 32, 16, 4, 7
    plan, ar, d = _make_plan(L, lib, sched, 0, nl, 0, monkeypatch, T, B, H, E, A, U, hetero=True)
    try:
        assert lib.parrot_decoder_backward_tick(plan) == 8  # (round 6: also the reference's own depth, three layers)
        recs = _trace(lib, plan, 1)
        bwd_once = {"dp"} | {f"dG{l}" for l in range(nl)} | {f"dC{l}" for l in range(nl)}
        acc = {"dw", "dw0", "dw_b", "dw0_b", "dw_c", "dw0_c"} | {f"dh{l}" for l in range(nl)} | {f"dh_b{l}" for l in range(nl)}
        acc |= {f"dhup{l}" for l in range(nl - 1)} | {f"dhup_b{l}" for l in range(nl - 1)} | {f"dhup_c{l}" for l in range(nl - 1)}
        n_bwd = _check(recs, ar, bwd_once, acc, T, {})
        assert n_bwd == 3 * (T + 2 * (nl - 1)), n_bwd  # three launches per tick, T + 2 (L - 1) ticks
        n = lib.parrot_decoder_trace_jobs(plan, 1, None, 0)
        buf = (C.c_longlong * (6 * n))()
        lib.parrot_decoder_trace_jobs(plan, 1, buf, n)
        jobs = [tuple(buf[6 * i:6 * i + 6]) for i in range(n)]  # (launch, job, M, N, Ksum, epi)
        assert max(j[4] for j in jobs if j[5] >= 0) == H  # no step-GEMM job walks more than K = H
        att_launches = {j[0] for j in jobs if j[5] == -2}
        assert any(j[0] in att_launches and j[5] >= 0 for j in jobs)  # GEMM work rides beside the attention backward
        for name in ["dw_b", "dw_c", "dw0_b", "dw0_c"] + [f"dh_b{l}" for l in range(nl)] + \
                [f"dhup_{x}{l}" for x in "bc" for l in range(nl - 1)]:
            kinds = {r[2] for r in recs if _owner(ar, r[3]) == name}
            assert (1 in kinds or 2 in kinds) and 0 in kinds, (name, kinds)
    finally:
        lib.parrot_decoder_destroy(plan)
    monkeypatch.setenv("PARROT_BWD_HETERO", "0")  # opt-out: the three-launch tick of bwd()
    plan, ar, d = _make_plan(L, lib, sched, 0, nl, 0, monkeypatch, T, B, H, E, A, U, hetero=True)
    try:
        assert lib.parrot_decoder_backward_tick(plan) == 0
        n = lib.parrot_decoder_trace_jobs(plan, 1, None, 0)
        buf = (C.c_longlong * (6 * n))()
        lib.parrot_decoder_trace_jobs(plan, 1, buf, n)
        assert max(buf[6 * i + 4] for i in range(n) if buf[6 * i + 5] >= 0) == 2 * H
    finally:
        lib.parrot_decoder_destroy(plan)


def test_tracer_sees_a_broken_order(monkeypatch):
    """The checker is not vacuous: the same trace with two consecutive launches swapped fails it."""
    L, lib = _lib()
    T, B, H, E, A, U = 4, 20, 32, 16, 4, 7
    plan, ar, d = _make_plan(L, lib, 5, 0, 2, 0, monkeypatch, T, B, H, E, A, U)
    try:
        recs = _trace(lib, plan, 0)
        f = 4
        slot = {"w": B * E * f, "kappa": B * A * f, "h0": B * H * f, "h1": B * H * f}
        once = {"w", "kappa", "a", "b", "phi", "att_sup", "h0", "h1", "z0", "r0", "rh0", "c0", "z1", "r1", "rh1", "c1",
                "seq_g1", "seq_c1"}
        _check(recs, ar, once, set(), T, slot)
        swapped = [((1 if r[0] == 0 else 0 if r[0] == 1 else r[0]),) + r[1:] for r in recs]  # candidate before gates
        with pytest.raises(AssertionError):
            _check(swapped, ar, once, set(), T, slot)
    finally:
        lib.parrot_decoder_destroy(plan)


def test_tick_model_prices_the_headline_plan(monkeypatch, capsys):
    """tools/tick_model.py (launch cost model of DESIGN.md 3.2 applied to a dry-run plan) runs without a GPU and lands on
    the measured scan times of BASELINE configs[1]: forward 24.8 ms on schedule 5, backward 26.7 ms on the K-balanced tick
    (bwd8; 28.7 ms on the three-launch tick, `--no-accumulators`), each within 10 %."""
    import importlib.util
    import re
    import sys
    _lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tick_model", os.path.join(root, "tools", "tick_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["tick_model.py", "--schedule", "5"])
    monkeypatch.setenv("PARROT_SCHEDULE", "5")  # (tick_model sets it itself; monkeypatch restores the environment)
    monkeypatch.setenv("PARROT_TRACE_ONLY", "1")
    mod.main()
    out = capsys.readouterr().out
    fwd = float(re.search(r"forward: \d+ launches, predicted ([0-9.]+) ms", out).group(1))
    bwd = float(re.search(r"backward: \d+ launches, predicted ([0-9.]+) ms", out).group(1))
    assert abs(fwd - 24.8) / 24.8 < 0.10 and abs(bwd - 26.7) / 26.7 < 0.10, out
    monkeypatch.setattr(sys, "argv", ["tick_model.py", "--schedule", "5", "--no-accumulators"])
    mod.main()
    out = capsys.readouterr().out
    bwd3 = float(re.search(r"backward: \d+ launches, predicted ([0-9.]+) ms", out).group(1))
    assert abs(bwd3 - 28.7) / 28.7 < 0.10, out
