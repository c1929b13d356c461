"""Prices the launches of a training-scan plan with the launch cost model measured in round 3 (DESIGN.md 3.2,
profiles/r03_launch_cost_model.txt) -- without a GPU: the plan is created on fake device addresses and dry-run through
parrot_decoder_trace_jobs.

    python tools/tick_model.py [--schedule 0|5|6] [--L 2] [--H 1024] [--B 64] [--T 800] [--cell gru|lstm]

Model (f32 step kernels, 32 x 32 tiles when the launch has >= 224 workgroups of them, else 32 x 16 tiles):
    launch = FIXED + SLOPE(tile) * Kmax / 1024 * rounds,   FIXED = 4.7 us, SLOPE(32x32) = 4.8 us, SLOPE(32x16) = 2.9 us,
    rounds = ceil(workgroups / 256) (one workgroup per CU; co-resident pairs gain <= 15 %, ignored),
    a launch that carries the attention step is at least ATT_FWD = 9.4 us (alone) / 12.1 us (beside GEMM workgroups),
    attention backward + state backward = 11.1 us.
Prints the predicted time of every distinct launch shape of a steady-state tick and the predicted scan times, next to
the measured ones where profiles/ has them."""
import argparse
import collections
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FIXED, SLOPE22, SLOPE21 = 4.7, 4.8, 2.9
ATT_ALONE, ATT_HETERO, ATT_BWD = 9.4, 12.1, 11.1


def fake_plan(L_, lib, a):
    os.environ["PARROT_SCHEDULE"] = str(a.schedule)
    os.environ["PARROT_TRACE_ONLY"] = "1"
    top = [0x10000000]

    def take(n):
        lo = top[0]
        top[0] = (lo + max(n, 16) + 4095) // 4096 * 4096 + 4096
        return lo
    d = L_.DecoderDesc()
    T, B, H, E, A, U = a.T, a.B, a.H, 256, 10, 200
    d.T, d.B, d.H, d.E, d.A, d.U, d.L = T, B, H, E, A, U, a.L
    cell = 1 if a.cell == "lstm" else 0
    d.cell, d.use_graph = cell, 0
    d.eps, d.alignment, d.sharpening, d.timing = 1e-5, 1.0, 1.0, 1.0
    gw = 4 * H if cell else 2 * H
    for l in range(a.L):
        K = H + E + l * H
        for n in ("Wg", "Wg_f", "Wg_r"):
            getattr(d, n)[l] = take(K * gw * 4)
        d.bg[l] = take(gw * 4)
        d.h[l], d.dh[l] = take((T + 1) * B * H * 4), take((T + 1) * B * H * 4)
        d.dG[l] = take(T * B * gw * 4)
        if l < a.L - 1:
            d.dhup[l] = take((T + 1) * B * H * 4)
        if not a.no_accumulators:  # second / third accumulators: the K-split backward ticks (LSTM bf16; bwd8 for 2 GRU layers)
            d.dh_b[l] = take((T + 1) * B * H * 4)
            if l < a.L - 1:
                d.dhup_b[l], d.dhup_c[l] = take((T + 1) * B * H * 4), take((T + 1) * B * H * 4)
        if l >= 1:
            d.seq_g[l] = take(T * B * gw * 4)
        if not cell:
            for n in ("Wc", "Wc_f", "Wc_r"):
                getattr(d, n)[l] = take(K * H * 4)
            d.bc[l] = take(H * 4)
            for n in ("z", "r", "rh", "c", "dC"):
                getattr(d, n)[l] = take(T * B * H * 4)
            if l >= 1:
                d.seq_c[l] = take(T * B * H * 4)
        else:
            d.cst[l], d.gate4[l], d.dcell[l] = take((T + 1) * B * H * 4), take(T * B * 4 * H * 4), take(B * H * 4)
    d.WattT, d.batt, d.ctx = take(3 * A * H * 4), take(3 * A * 4), take(B * U * E * 4)
    d.w, d.kappa = take((T + 1) * B * E * 4), take((T + 1) * B * A * 4)
    d.a, d.b, d.phi = take(T * B * A * 4), take(T * B * A * 4), take(T * B * U * 4)
    d.dw, d.dw0, d.dkappa = take((T + 1) * B * E * 4), take((T + 1) * B * E * 4), take(B * A * 4)
    d.dp, d.att_sup = take(T * B * 3 * A * 4), take(T * B * 8)
    if not a.no_accumulators:
        d.dw_b, d.dw_c = take((T + 1) * B * E * 4), take((T + 1) * B * E * 4)
        d.dw0_b, d.dw0_c = take((T + 1) * B * E * 4), take((T + 1) * B * E * 4)
    plan = C.c_void_p()
    rc = lib.parrot_decoder_create(C.byref(d), C.byref(plan))
    assert rc == 0, rc
    return plan


def jobs_of(lib, plan, which):
    n = lib.parrot_decoder_trace_jobs(plan, which, None, 0)
    assert n > 0, n
    buf = (C.c_longlong * (6 * n))()
    lib.parrot_decoder_trace_jobs(plan, which, buf, n)
    by = collections.defaultdict(list)
    for i in range(n):
        launch, job, M, N, K, epi = buf[6 * i:6 * i + 6]
        by[launch].append((job, M, N, K, epi))
    return [by[k] for k in sorted(by)]


def price(jobs, cell):
    """(us, description) of one launch."""
    gemm = [j for j in jobs if j[4] >= 0]
    att = [j for j in jobs if j[4] < 0]
    if not gemm:
        return (ATT_BWD if att and att[0][4] == -2 else ATT_ALONE), "attention"
    rows = max(math.ceil(j[1] / 32) for j in gemm)
    tiles16 = sum((j[2] + 15) // 16 for j in gemm)               # 16-column tiles
    wg22 = sum(math.ceil(math.ceil(j[2] / 16) / 2) for j in gemm) * rows
    use22 = wg22 >= 224 or (att and wg22 >= 160)
    wgs = wg22 if use22 else tiles16 * rows
    kmax = max(j[3] for j in gemm)
    rounds = math.ceil((wgs + (64 if att and att[0][4] == -1 else 0)) / 256)
    t = FIXED + (SLOPE22 if use22 else SLOPE21) * kmax / 1024 * rounds
    what = f"{len(gemm)} jobs, {wgs} workgroups of 32x{'32' if use22 else '16'}, Kmax {kmax}" + (f", {rounds} rounds" if rounds > 1 else "")
    if att and att[0][4] == -1:
        t = max(t, ATT_HETERO)
        what += " + attention"
    elif att:
        # bwd8's attention launch: 64 attention rows + 64 state rows of the upper layer lead the grid (1024-thread blocks,
        # a CU each); the CUs of the short state rows (~3 us) then take the GEMM workgroups that found no free CU
        t = max(ATT_HETERO, t + (3.0 if wgs + 80 > 256 else 0.0))  # (64 attention rows + 16 blocks of 4 state rows)
        what += " + attention backward"
    return t, what


def whatif(L=2, H=1024, E=256, B=64, T=800):
    """Round 4: prices two re-arrangements of the BACKWARD tick against today's, from the same
    measured constants, before anything is built.  Work unit = one 32 x 32 output tile over K = 1024 (4.8 us of CU time)."""
    rows = math.ceil(B / 32)
    cols = lambda n: math.ceil(n / 32) * rows            # 32 x 32 tiles of an [B, n] output
    # products of one backward tick (GRU): per layer l, on the chain: d(rh) = dC.Wc[0:H]^T (K = H) and
    # dh_prev += dG.Wg[0:H]^T (K = 2H); towards the attention: dw (+)= dC.Wc[H:H+E]^T, dG.Wg[H:H+E]^T; towards every
    # lower layer p < l: dhup_p (+)= dC.Wc[..]^T, dG.Wg[..]^T
    kH = H / 1024.0
    chain_x = sum(cols(H) for l in range(L))                              # d(rh), K = H
    chain_y = sum(cols(H) * 2 for l in range(L))                          # dG -> dh_prev, K = 2H = two K = H units
    dw = sum((cols(E)) * 3 for l in range(L))                             # dC part + the two dG halves
    down = sum(cols(H) * 3 * l for l in range(L))                         # the same three parts per lower layer
    units = (chain_x + chain_y + dw + down)
    print(f"backward tick, L={L} H={H} B={B}: {units} units of 32x32xK{H} "
          f"({chain_x} d(rh) + {chain_y} dG->dh_prev on the chain, {dw} dw, {down} downward) = {units / 256:.2f} rounds of 256")
    unit_us = SLOPE22 * kH
    today = ATT_BWD + (FIXED + unit_us) + (FIXED + 2 * unit_us)
    print(f"  today (S | X: K=H | Y: K=2H, 224 workgroups each): {ATT_BWD} + {FIXED + unit_us:.1f} + {FIXED + 2 * unit_us:.1f} "
          f"= {today:.1f} us/tick -> {today * T / 1000:.1f} ms   (measured 28.7 ms)")
    # (a) K-balanced launches: every K = 2H product cut into its z and r halves writing separate buffers (the consumer adds
    #     them); X = d(rh) + dG_z->dh_prev (both only need the state backward), Y = dG_r->dh_prev + dw0 + the dC share of the
    #     downward products, and -- with layer l two ticks ahead of layer l-1 -- the other downward products ride in the NEXT
    #     tick's attention launch as GEMM workgroups beside the 64 attention blocks (heterogeneous, as ska_kernel forward).
    x_a = chain_x + chain_y // 2
    y_a = chain_y // 2 + (dw // L) + (cols(H) + cols(E)) * (L - 1)
    s_a = units - x_a - y_a
    ok = x_a <= 256 and y_a <= 256 and s_a + 64 <= 256
    t_a = max(ATT_HETERO, FIXED + unit_us) + (FIXED + unit_us) * 2
    print(f"  (a) K-balanced launches + heterogeneous attention launch: X {x_a} / Y {y_a} / S {s_a}+64 workgroups "
          f"({'all <= 256' if ok else 'DOES NOT FIT one round'}), every K = H: {max(ATT_HETERO, FIXED + unit_us):.1f} + "
          f"{FIXED + unit_us:.1f} + {FIXED + unit_us:.1f} = {t_a:.1f} us/tick -> {t_a * T / 1000:.1f} ms "
          f"({(today - t_a) * T / 1000:+.1f} ms)")
    # (b) resident workgroups pulling units from per-XCD queues.  Measured ingredients (DESIGN 3.4,
    #     profiles/r02_persist_*): a unit carries ~3.5 us of fixed cost next to its K loop (descriptor + first operand round
    #     trip 1.5-2 + split-K reduce 0.5 + write-through drain 1.2); a dependent hand-off between CUs on different XCDs =
    #     producer drain 1.2 + sc1 visibility round trip 1.5 + dequeue 0.3-1.3 (guide: sharded dequeue idle / streaming).
    UNIT_FIXED, HANDOFF = 3.5, 1.2 + 1.5 + 0.8
    link = HANDOFF + 1.0 + unit_us                      # hand-off + first operand fetch + K loop of a K = H unit
    chain_b = ATT_BWD + 2 * link                        # attention/state chain, then d(rh), then the dG_r half of dh_prev
    fill_b = units * (unit_us + UNIT_FIXED) / 256       # all units packed perfectly on 256 CUs
    t_b = max(chain_b, fill_b)
    print(f"  (b) queue-fed units: dependent chain {ATT_BWD} + 2 x {link:.1f} = {chain_b:.1f} us, perfectly packed work "
          f"{fill_b:.1f} us -> {t_b:.1f} us/tick -> {t_b * T / 1000:.1f} ms ({(today - t_b) * T / 1000:+.1f} ms); the "
          f"forward tick is chain-bound already (10.7 + 8.3 + 12.1 = sum of its three dependent launches): no gain there")
    print(f"  verdict: (b) predicts {(today - t_b) * T / 1000:.1f} ms AT BEST -- every hand-off at its measured minimum and the "
          f"{units} units packed perfectly -- i.e. at the 4 ms bar, not above it; the one resident design built so far (the "
          f"round-2 phase machine, same hand-off costs) only matched the launches.  (a) gets {(today - t_a) * T / 1000:.1f} ms "
          f"of it inside the launch design, at the price of a second heterogeneous kernel (attention BACKWARD blocks beside "
          f"GEMM workgroups), a two-tick layer skew and three partial buffers per cross-layer gradient.")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--whatif", action="store_true", help="price the round-4 re-arrangements of the backward tick")
    ap.add_argument("--schedule", type=int, default=5)
    ap.add_argument("--L", type=int, default=2)
    ap.add_argument("--H", type=int, default=1024)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--T", type=int, default=800)
    ap.add_argument("--cell", default="gru")
    ap.add_argument("--no-accumulators", action="store_true",
                    help="leave the second / third gradient accumulators out of the descriptor: the three-launch backward tick of rounds 1-3")
    a = ap.parse_args()
    if a.whatif:
        return whatif(L=a.L, H=a.H, B=a.B, T=a.T)
    from parrot_amd import _lib as L_
    lib = L_.load()
    plan = fake_plan(L_, lib, a)
    print(f"schedule {lib.parrot_decoder_schedule(plan)} (asked {a.schedule}), L={a.L} H={a.H} B={a.B} T={a.T} {a.cell}")
    for which, name in ((0, "forward"), (1, "backward")):
        launches = jobs_of(lib, plan, which)
        total = 0.0
        shapes = collections.Counter()
        for jobs in launches:
            t, what = price(jobs, a.cell)
            total += t
            shapes[(round(t, 1), what)] += 1
        print(f"  {name}: {len(launches)} launches, predicted {total / 1000:.1f} ms")
        for (t, what), n in shapes.most_common(6):
            print(f"      {n:5d} x {t:5.1f} us  {what}")
    lib.parrot_decoder_destroy(plan)


if __name__ == "__main__":
    main()
