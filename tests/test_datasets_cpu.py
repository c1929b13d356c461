"""Host-side batch producer semantics (reference datasets.py) -- CPU only."""
import numpy as np

from oracle import quantize_ref as Q
from parrot_amd.datasets import SegmentSequence, _chunk, get_raw_transformer, parrot_stream


def test_segment_sequence_tbptt_windows():
    """datasets.py:66-138, 286-292: windows of seq_size+1 with hop seq_size (overlap 1), trailing
    partial window dropped (return_last=False -> min_size = 10 + window), flag 1 on the first window."""
    T, B = 137, 3
    feats = np.arange(T * B, dtype='float32').reshape(T, B)
    lab = np.zeros((B, 5), dtype='int32')
    seq_size = 50
    seg = SegmentSequence([(feats, lab)], ('features', 'labels'), seq_size=seq_size + 1, share_value=1,
                          return_last=False, add_flag=True, which_sources=('features',))
    out = list(seg)
    assert seg.sources == ('features', 'labels', 'start_flag')
    # step sequence: 0, 50, 100 ... stop when step + (10 + 51) >= 137  -> after the window at 50 (100+61>=137)
    assert [o[0].shape[0] for o in out] == [51, 51]
    assert [o[2] for o in out] == [1, 0]
    assert np.array_equal(out[0][0][-1], out[1][0][0])  # one-frame overlap (teacher-forcing shift)
    assert out[0][1] is lab  # unsegmented sources pass through
    # a batch shorter than one window still yields its first (short) window
    short = SegmentSequence([(feats[:30], lab)], ('features', 'labels'), seq_size=51, share_value=1,
                            return_last=False, add_flag=True, which_sources=('features',))
    o = list(short)
    assert len(o) == 1 and o[0][0].shape[0] == 30 and o[0][2] == 1


def test_stream_layout_and_sorting():
    s = parrot_stream('vctk', which_sets=('train',), batch_size=4, seq_size=20, labels_type='text',
                      raw_data=False, num_examples=10, sorting_mult=2, use_speaker=True)
    assert s.sources == ('features', 'features_mask', 'labels', 'labels_mask', 'speaker_index', 'start_flag')
    items = list(s.get_epoch_iterator())
    assert len(items) > 0
    for f, fm, lab, lm, spk, flag in items:
        assert f.ndim == 3 and f.shape[1] == 4 and f.shape[2] == 63 and f.dtype == np.float32  # time-major
        assert fm.shape == f.shape[:2] and lab.shape == lm.shape and lab.shape[0] == 4
        assert spk.shape == (4, 1) and flag in (0, 1)
    # 10 examples, sorting window 8 -> batches of 4,4 (+ a dropped batch of 2): 2 padded batches
    assert sum(i[-1] for i in items) == 2
    # epochs reshuffle for the train set but are deterministic per construction
    a = [i[0].shape for i in parrot_stream('vctk', batch_size=4, seq_size=20, labels_type='text', raw_data=False,
                                           num_examples=10, sorting_mult=2).get_epoch_iterator()]
    b = [i[0].shape for i in parrot_stream('vctk', batch_size=4, seq_size=20, labels_type='text', raw_data=False,
                                           num_examples=10, sorting_mult=2).get_epoch_iterator()]
    assert a == b


def test_raw_audio_chunk_and_quantise_with_oracle_quantiser():
    """datasets.py:194-203 with the oracle quantiser injected (the product default is the HIP kernel)."""
    rng = np.random.RandomState(0)
    B, T = 3, 7
    raw = rng.randn(B, T * 80).astype('float32')
    chunks = _chunk(raw)  # [T,B,80]
    assert chunks.shape == (T, B, 80) and np.array_equal(chunks[2, 1], raw[1, 160:240])
    tf = get_raw_transformer('mu-law', 256, quantizer=lambda x, ql, qt: Q.batch_quantize(x, ql, qt))
    q = tf(chunks)
    assert q.shape == (T, B, 80) and q.dtype == np.int16
    ref = Q.batch_quantize(raw, 256, 'mu-law')
    assert np.array_equal(q.transpose(1, 0, 2).reshape(B, -1), ref)


def test_end_of_utterance_heuristic():
    from parrot_amd.utils import end_of_utterance
    S, U = 30, 6
    phi = np.zeros((S, U + 1), dtype='float32')
    for t in range(S):
        phi[t, min(U, t // 4)] = 1.0  # the window walks over the text, reaches the end at t = 24
    assert end_of_utterance(phi, U, 100) == 24 + 40
    assert end_of_utterance(phi, U, 50) == 50
    assert end_of_utterance(phi[:10], U, 77) == 77  # never reaches the end


def _datasets_golden():
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("mk_ds", os.path.join(here, "golden", "make_datasets_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk, np.load(os.path.join(here, "golden", "datasets_golden.npz"))


def test_segment_sequence_matches_the_reference_class():
    """tests/golden/datasets_golden.npz: the windows the reference's own SegmentSequence.get_data (datasets.py:109-138,
    executed by make_datasets_golden.py) cut out of seeded padded batches -- two batches in a row per case, so the reset
    between batches is covered -- for the training configuration and for overlap / return_last variants."""
    mk, gold = _datasets_golden()
    for case, (n, seq, share, ret_last, min_size) in mk.CASES.items():
        batch = mk.batch_for(case)
        seg = SegmentSequence([batch, batch], ('features', 'features_mask', 'labels'), seq_size=seq, share_value=share,
                              return_last=ret_last, add_flag=True, min_size=min_size,
                              which_sources=('features', 'features_mask'))
        out = list(seg)
        assert len(out) == int(gold[f'{case}|n']), case
        for k, (f, fm, lab, flag) in enumerate(out):
            assert np.array_equal(f, gold[f'{case}|{k}|features']), (case, k)
            assert np.array_equal(fm, gold[f'{case}|{k}|features_mask']), (case, k)
            assert np.array_equal(lab, gold[f'{case}|{k}|labels']), (case, k)
            assert int(flag) == int(gold[f'{case}|{k}|flag']), (case, k)


def test_raw_transformer_matches_the_reference_function():
    """_chunk + get_raw_transformer (datasets.py:28-29, 187-196) as executed from the reference file on the reference's
    quantize.py: same frames, same integer codes (the oracle quantiser stands in for the HIP kernel on CPU; the kernel is
    bit-exact against the same module, the quantiser tests)."""
    mk, gold = _datasets_golden()
    chunked = _chunk(gold['raw|in'])
    assert np.array_equal(chunked, gold['raw|chunked'])
    for q_type in ('mu-law', 'linear'):
        tf = get_raw_transformer(q_type, 256, quantizer=lambda x, ql, qt: Q.batch_quantize(x, ql, qt))
        got = tf(chunked.copy())
        assert got.shape == gold[f'raw|{q_type}'].shape
        assert np.array_equal(got.astype(np.int64), gold[f'raw|{q_type}'].astype(np.int64)), q_type


def test_parrot_stream_matches_the_reference_pipeline():
    """tests/golden/stream_golden.npz: every item the reference's own parrot_stream (datasets.py:199-298, executed from the
    reference file by make_stream_golden.py on restated Fuel classes) yields over a 19-example synthetic validation set --
    sort window 8, batches of 4 (the short last batch dropped), padding + masks, time-major transpose, 80-sample chunks
    quantised per utterance, TBPTT windows of 21 frames with one frame of overlap, start flag, noise source."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("mk_st", os.path.join(here, "golden", "make_stream_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = np.load(os.path.join(here, "golden", "stream_golden.npz"))
    ds = mk.synthetic_examples()
    for case, kw in mk.CASES.items():
        s = parrot_stream('vctk', dataset=ds, num_examples=mk.N_EXAMPLES,
                          quantizer=lambda x, ql, qt: Q.batch_quantize(x, ql, qt), **dict(mk.COMMON, **kw))
        assert sorted(s.sources) == gold[f'{case}|sources'].tolist(), (case, s.sources)
        items = list(s.get_epoch_iterator())
        assert len(items) == int(gold[f'{case}|n']), (case, len(items))
        for k, item in enumerate(items):
            for name, val in zip(s.sources, item):
                ref = gold[f'{case}|{k}|{name}']
                got = np.asarray(val)
                assert got.shape == ref.shape, (case, k, name, got.shape, ref.shape)
                if name == 'raw_audio':
                    assert np.array_equal(got.astype(np.int64), ref.astype(np.int64)), (case, k, name)
                else:
                    assert np.array_equal(got, ref), (case, k, name)


def test_async_loader_propagates_worker_errors_and_stops_when_the_consumer_leaves():
    """PinnedAsyncLoader (the replacement of Fuel's server/iterator pair, datasets.py:199-298): an exception raised
    while a batch is produced reaches the consumer (it is not an end of epoch), and a consumer that breaks out early
    leaves no producer thread blocked on the queue."""
    import threading

    import numpy as np
    import pytest

    from parrot_amd.datasets import PinnedAsyncLoader

    class Stream:
        sources = ('x',)

        def __init__(self, n, fail_at=None):
            self.n, self.fail_at = n, fail_at

        def get_epoch_iterator(self):
            for i in range(self.n):
                if i == self.fail_at:
                    raise ValueError("quantiser blew up")
                yield (np.full((2, 3), i, dtype=np.float32),)

    got = [int(b['x'][0, 0]) for b in PinnedAsyncLoader(Stream(5), 'cpu')]
    assert got == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError, match="quantiser"):
        for _ in PinnedAsyncLoader(Stream(5, fail_at=2), 'cpu'):
            pass
    before = threading.active_count()
    for i, _ in enumerate(PinnedAsyncLoader(Stream(1000), 'cpu', depth=2)):
        if i == 1:
            break
    import time
    for _ in range(50):
        if threading.active_count() <= before:
            break
        time.sleep(0.1)
    assert threading.active_count() <= before, "the producer thread is still alive after the consumer left"
