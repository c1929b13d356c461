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
    if att:
        t = max(t, ATT_HETERO if att[0][4] == -1 else ATT_BWD)
        what += " + attention"
    return t, what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--schedule", type=int, default=5)
    ap.add_argument("--L", type=int, default=2)
    ap.add_argument("--H", type=int, default=1024)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--T", type=int, default=800)
    ap.add_argument("--cell", default="gru")
    a = ap.parse_args()
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
