"""Host-side training logic that needs no GPU: the LearningRateSchedule arithmetic (extensions.py:83-152), the
data-parallel batch producer (shards of the same global batches) and the bounded workspace cache."""
import numpy as np


class _FakeTrainer:
    def __init__(self):
        self.lr, self.cuts = 1.0, 0

    def cut_learning_rate(self, factor=0.5):
        self.lr *= factor
        self.cuts += 1


def test_learning_rate_schedule_patience_cuts_and_stop():
    from parrot_amd.trainer import LearningRateSchedule
    tr, reloads = _FakeTrainer(), []
    s = LearningRateSchedule(tr, lambda: reloads.append(1), patience=3, num_cuts=2, cut_size=.5)
    assert s.update(None) == (False, False)          # no validation value logged yet
    assert s.update(5.0) == (False, False)           # new best
    assert s.update(4.0) == (False, False)           # new best, counter reset
    assert s.update(4.5) == (False, False)           # 1
    assert s.update(4.0) == (False, False)           # 2 (not strictly better)
    assert s.update(4.1) == (True, False)            # 3 = patience -> cut 1: reload best, halve, zero buffers
    assert tr.lr == 0.5 and len(reloads) == 1 and s.counter == 0
    assert s.update(float('nan')) == (True, True)    # NaN skips the remaining patience (extensions.py:126-128)
    assert tr.lr == 0.25 and s.count_cuts == 2 and len(reloads) == 2
    assert s.best_value == 4.0


def test_reference_wiring_patience10_cuts5():
    from parrot_amd.trainer import LearningRateSchedule
    tr = _FakeTrainer()
    s = LearningRateSchedule(tr, lambda: None, patience=10, num_cuts=5)  # train.py:175-182
    s.update(1.0)
    fin = False
    n = 0
    while not fin:
        _, fin = s.update(2.0)
        n += 1
    assert n == 50 and tr.cuts == 5 and abs(tr.lr - 0.5 ** 5) < 1e-12


def _collect(stream):
    return [dict(zip(stream.sources, item)) for item in stream.get_epoch_iterator()]


def test_sharded_producer_yields_the_shards_of_the_global_batches():
    """Every rank's stream == the matching rows of the single-process stream: same number of windows, same
    padding lengths, same start flags (so all ranks issue the same number of all-reduces)."""
    from parrot_amd.datasets import VoiceData, parrot_stream
    from parrot_amd.dist import shard_batch
    ds = VoiceData('vctk', ('train',), num_examples=24, seed=3)
    kw = dict(batch_size=6, seq_size=40, labels_type='text', raw_data=False, dataset=ds, use_speaker=True, seed=11)
    full = _collect(parrot_stream('vctk', **kw))
    assert len(full) > 3
    world = 3
    for rank in range(world):
        part = _collect(parrot_stream('vctk', shard=(rank, world), **kw))
        lo, hi = shard_batch(6, rank, world)
        assert len(part) == len(full)
        for a, b in zip(part, full):
            assert a['start_flag'] == b['start_flag']
            assert np.array_equal(a['features'], b['features'][:, lo:hi])
            assert np.array_equal(a['features_mask'], b['features_mask'][:, lo:hi])
            assert np.array_equal(a['labels'], b['labels'][lo:hi])
            assert np.array_equal(a['labels_mask'], b['labels_mask'][lo:hi])
            assert np.array_equal(a['speaker_index'], b['speaker_index'][lo:hi])


def test_workspace_cache_is_bounded_and_frees_evicted_entries():
    from parrot_amd.model import _LRU
    freed = []
    c = _LRU(3, freed.append)
    for k in range(10):
        c[('dec', 50 + k, 64, 200)] = k
        c.get(('dec', 50, 64, 200))  # the shape in use stays
    assert len(c) == 3 and ('dec', 50, 64, 200) in c
    assert freed == [1, 2, 3, 4, 5, 6, 7]
    c.clear()
    assert len(c) == 0 and len(freed) == 10


def test_learning_rate_schedule_matches_the_reference_class():
    """tests/golden/schedule_golden.json: what the reference's own LearningRateSchedule.do (extensions.py:119-152, executed
    by make_schedule_golden.py on stand-ins for its Blocks collaborators) did at every validation check of four seeded
    trajectories -- learning rate, number of cuts, patience counter, best-parameter reload + buffer reset, finish."""
    import json
    import os
    from parrot_amd.trainer import LearningRateSchedule
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "schedule_golden.json")))
    assert len(gold) == 8
    for key, case in gold.items():
        _, p, c = key.split('|')
        tr, reloads = _FakeTrainer(), []
        tr.lr = 1e-4
        s = LearningRateSchedule(tr, lambda: reloads.append(1), patience=int(p[1:]), num_cuts=int(c[1:]), cut_size=.5)
        finished = False
        for v, row in zip(case['values'], case['rows']):
            assert not finished, key
            n_rel, n_cut = len(reloads), tr.cuts
            cut, fin = s.update(float('nan') if v is None else v)
            finished = finished or fin
            assert abs(tr.lr - row['lr']) <= 1e-12 * row['lr'], (key, tr.lr, row['lr'])
            assert s.count_cuts == row['cuts'] and s.counter == row['counter'], (key, s.count_cuts, s.counter, row)
            assert cut == row['reloaded'] == row['buffers_zeroed'], (key, cut, row)
            assert (len(reloads) - n_rel == 1) == cut and (tr.cuts - n_cut == 1) == cut
            assert finished == row['finish'], (key, finished, row)
        assert finished == case['rows'][-1]['finish']
