"""The decode machine's step cut along K by the age of its operands (plans_decode.hip, build_persist_pieces): planned and
replayed symbolically on the CPU -- no device memory is touched (parrot_sample_plan_pieces_dry)."""
import ctypes as C

import pytest

from parrot_amd import _lib


def _desc(L=2, H=1024, E=512, B=16, S=1000, R=1024, fb=(0,), speaker=False, fbc=False, attfold=False):
    d = _lib.SampleDesc()
    d.S, d.B, d.H, d.E, d.A, d.U, d.L, d.O, d.R, d.ldx = S, B, H, E, 10, 100, L, 63, R, 64
    fake = 0x7000_0000_0000  # never dereferenced by the dry run
    for l in range(L):
        d.Wg_t[l], d.Wc_t[l] = fake, fake
        d.bg[l], d.bc[l] = fake, fake
        if l in fb:
            d.Wfg[l], d.Wfc[l] = fake, fake
        if speaker:
            d.seq_g[l], d.seq_c[l] = fake, fake
    d.Wro_t, d.ro_const, d.x = fake, fake, fake
    if fbc:  # layer 0's matrices with the composed feedback rows appended (ParrotSampleDesc::Wgx_t / Wcx_t, round 5)
        d.Wgx_t[0], d.Wcx_t[0] = fake, fake
    if attfold:  # the attention projection as a fragment-major [H, 32] matrix (ParrotSampleDesc::Watt_t, round 5)
        d.Watt_t = fake
    return d


def _plan(d, nwg=256):
    info = (C.c_int * 16)()
    rc = _lib.load().parrot_sample_plan_pieces_dry(C.byref(d), nwg, info)
    return rc, list(info)


def test_configs2_plan_is_legal_and_fits_one_unit_per_workgroup():
    rc, info = _plan(_desc())
    assert rc == 0 and info[2] == 0
    assert info[0] == 6                      # 2L + 2 phases per step (7 with whole-K products)
    assert info[1] == 10                     # partial-sum buffers: G0 2, C0 2, G1 2, C1 2, output 2
    assert all(0 < n <= 256 for n in info[4:10]), info
    assert sum(info[4:10]) == info[3]


def test_fed_back_frame_out_of_the_chain_plan_is_legal():
    """Round 5 (Wgx_t / Wcx_t given, weak feedback, L >= 2): 2L + 1 phases, the output product beside the next step's gate
    phase, x_pre as a group of its own; the symbolic replay passes; other feedback patterns / L = 1 keep 2L + 2 phases."""
    rc, info = _plan(_desc(fbc=True))
    assert rc == 0 and info[2] == 0 and info[15] == 1, info
    assert info[0] == 5
    assert all(0 < n <= 256 for n in info[4:9]), info
    assert sum(info[4:9]) == info[3]
    for L, fb in ((3, (0,)), (2, (0,))):
        rc, info = _plan(_desc(L=L, H=256, E=128, B=16, S=50, R=256, fb=fb, fbc=True))
        assert rc == 0 and info[2] == 0 and info[15] == 1 and info[0] == 2 * L + 1, info
    for L, fb in ((1, (0,)), (2, (0, 1)), (2, ())):  # not the pattern the composition covers: the 2L + 2 phases
        rc, info = _plan(_desc(L=L, H=256, E=128, B=16, S=50, R=256, fb=fb, fbc=True))
        assert rc == 0 and info[2] == 0 and info[15] == 0 and info[0] == 2 * L + 2, info


def test_attention_projection_fold_is_planned_for_small_batches_only():
    """Watt_t given: layer 0's candidate units publish the projection's partial sums (checked by the symbolic replay: the
    attention row reads them a phase later); B > 16 (more than one row block per unit) keeps the row's own projection."""
    for fbc in (False, True):
        rc, info = _plan(_desc(fbc=fbc, attfold=True))
        assert rc == 0 and info[2] == 0 and info[15] == (3 if fbc else 2), info
    rc, info = _plan(_desc(B=32, fbc=True, attfold=True))
    assert rc == 0 and info[2] == 0 and info[15] == 1, info
    rc, info = _plan(_desc(L=1, H=256, E=128, S=20, R=256, attfold=True))
    assert rc == 0 and info[2] == 0 and info[15] == 2, info


def test_fbc_env_switch(monkeypatch):
    monkeypatch.setenv("PARROT_PM_FBC", "0")
    rc, info = _plan(_desc(fbc=True))
    assert rc == 0 and info[15] == 0 and info[0] == 6


@pytest.mark.parametrize("L,fb,speaker", [(1, (0,), False), (1, (), False), (2, (), False), (2, (0, 1), False),
                                          (2, (0,), True), (3, (0,), False), (3, (0, 1, 2), True)])
def test_plans_of_other_stacks_are_legal(L, fb, speaker):
    d = _desc(L=L, H=256, E=128, B=16, S=50, R=256, fb=fb, speaker=speaker)
    rc, info = _plan(d)
    assert rc == 0 and info[2] == 0, info
    assert info[0] == 2 * L + 2


def test_too_few_workgroups_is_refused_not_misplanned():
    rc, info = _plan(_desc(), nwg=100)       # 128 gate tiles per phase do not fit 100 workgroups
    assert rc != 0


def test_env_switch_keeps_the_whole_k_phases(monkeypatch):
    monkeypatch.setenv("PARROT_PM_PIECES", "0")
    rc, info = _plan(_desc())
    assert rc != 0


def test_composed_readout_output_is_the_same_affine_map():
    """model.compose_readout_output: XR . (Wr . Wo) + const == (XR . Wr + br + radd) . Wo + bo + oadd (model.py:992-1013),
    with and without the speaker terms, padding columns zero."""
    import torch
    from parrot_amd.model import compose_readout_output
    g = torch.Generator().manual_seed(3)
    K, R, O, N = 160, 48, 63, 5
    Wr, Wo = torch.randn(K, R, generator=g) / K ** 0.5, torch.randn(R, O, generator=g) / R ** 0.5
    br, bo = torch.randn(R, generator=g), torch.randn(O, generator=g)
    Wo_pad, bo_pad = torch.zeros(R, 64), torch.zeros(64)
    Wo_pad[:, :O], bo_pad[:O] = Wo, bo
    XR = torch.randn(N, K, generator=g)
    for speaker in (False, True):
        radd = torch.randn(N, R, generator=g) if speaker else None
        oadd_pad = None
        if speaker:
            oadd_pad = torch.zeros(N, 64)
            oadd_pad[:, :O] = torch.randn(N, O, generator=g)
        Wro, c = compose_readout_output(Wr, Wo_pad, br, radd, bo_pad, oadd_pad, N)
        assert Wro.shape == (K, 64) and c.shape == (N, 64) and Wro.dtype == torch.float32
        ro = XR.double() @ Wr.double() + br.double() + (radd.double() if speaker else 0)
        ref = ro @ Wo.double() + bo.double() + (oadd_pad[:, :O].double() if speaker else 0)
        got = XR.double() @ Wro.double() + c.double()
        assert torch.allclose(got[:, :O], ref, rtol=0, atol=2e-6)
        assert float(got[:, O:].abs().max()) == 0.0


def test_random_stacks_are_either_planned_legally_or_refused():
    """Whatever the shape: a returned plan passes the symbolic replay and fits one unit per workgroup and phase; anything
    else is refused (the caller then builds the whole-K phases or the per-step launches)."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=80, deadline=None, derandomize=True)
    @given(L=st.integers(1, 3), h=st.integers(1, 96), e=st.integers(1, 48), B=st.integers(1, 64),
           fbmask=st.integers(0, 7), speaker=st.booleans(), nwg=st.sampled_from([64, 128, 208, 256]), fbc=st.booleans(),
           attfold=st.booleans())
    def run(L, h, e, B, fbmask, speaker, nwg, fbc, attfold):
        fb = tuple(l for l in range(L) if fbmask >> l & 1)
        d = _desc(L=L, H=16 * h, E=16 * e, B=B, S=7, R=64, fb=fb, speaker=speaker, fbc=fbc, attfold=attfold)
        rc, info = _plan(d, nwg=nwg)
        if rc == 0:
            assert info[2] == 0 and info[0] == (2 * L + 1 if info[15] & 1 else 2 * L + 2)
            assert all(n <= nwg for n in info[4:4 + info[0]]), (info, nwg)
            assert sum(info[4:4 + info[0]]) == info[3]
        else:
            assert info[2] == 0, (rc, info)   # refused for size, never because the replay found an illegal table

    run()
