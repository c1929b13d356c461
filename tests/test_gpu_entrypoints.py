"""End-to-end entry points on the GPU: BASELINE configs[0] (1-layer GRU h=256, batch=4, 10 toy
utterances) through train.py, then sample.py on the saved parameters."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_then_sample_cfg1(dev, tmp_path, monkeypatch):
    monkeypatch.setenv("RESULTS_DIR", str(tmp_path))
    sys.path.insert(0, ROOT)
    import importlib
    train = importlib.import_module("train")
    sample = importlib.import_module("sample")
    argv = ["--experiment_name", "toy", "--rnn_h_dim", "256", "--readouts_dim", "256", "--batch_size", "4",
            "--seq_size", "50", "--num_layers", "1", "--labels_type", "text", "--synthetic_examples", "10",
            "--save_every", "4", "--max_steps", "8", "--save_dir", str(tmp_path), "--weak_feedback", "1"]
    train.main(argv)
    assert os.path.exists(os.path.join(str(tmp_path), "vctk", "pkl", "best_toy.tar"))
    gen_x, lengths = sample.main(["--experiment_name", "toy", "--num_samples", "4", "--num_steps", "60",
                                  "--save_dir", str(tmp_path)])
    assert gen_x.shape == (4, 60, 63) and np.isfinite(gen_x).all()
    assert len(lengths) == 4
    assert os.path.exists(os.path.join(str(tmp_path), "vctk", "samples", "best_sample_0.npy"))


def test_trainer_reduces_cost_and_pinned_loader(dev):
    from parrot_amd.datasets import PinnedAsyncLoader, parrot_stream
    from parrot_amd.model import Parrot
    from parrot_amd.trainer import Trainer
    m = Parrot(device=dev, num_layers=2, rnn_h_dim=128, readouts_dim=128, encoder_type='bidirectional',
               weak_feedback=True).initialize()
    tr = Trainer(m, learning_rate=1e-3)
    stream = parrot_stream('vctk', batch_size=4, seq_size=30, labels_type='text', raw_data=False, num_examples=8,
                           sorting_mult=1)
    costs = []
    for epoch in range(6):
        for b in PinnedAsyncLoader(stream, dev):
            assert b['features'].is_cuda
            costs.append(float(tr.step(b['features'], b['features_mask'], b['labels'], b['labels_mask'], None,
                                       b['start_flag'])))
    assert np.isfinite(costs).all()
    assert np.mean(costs[-4:]) < np.mean(costs[:4])
    m.close()


def test_raw_audio_stream_uses_hip_quantiser(dev):
    from oracle import quantize_ref as Q
    from parrot_amd.datasets import VoiceData, parrot_stream
    ds = VoiceData('vctk', ('train',), num_examples=4, seed=5)
    s = parrot_stream('vctk', batch_size=4, seq_size=500, labels_type='text', raw_data=True, dataset=ds,
                      sorting_mult=1)
    item = dict(zip(s.sources, next(s.get_epoch_iterator())))
    raw = item['raw_audio']
    assert raw.dtype == np.int16 and raw.shape[2] == 80 and raw.shape[1] == 4
    exs = sorted(ds.examples, key=lambda e: len(e['features']))
    n = max(len(e['raw_audio']) for e in exs)
    padded = np.zeros((4, n), dtype='float32')
    for i, e in enumerate(exs):
        padded[i, :len(e['raw_audio'])] = e['raw_audio']
    ref = Q.batch_quantize(padded, 256, 'mu-law')
    got = raw.transpose(1, 0, 2).reshape(4, -1)
    assert np.array_equal(got, ref[:, :got.shape[1]])


def test_train_raw_output_then_sample_raw(dev, tmp_path, monkeypatch):
    """--raw_output (train.py:36,46,76,92-93; sample.py:165-173): the raw-audio stream feeds the SampleRNN head inside
    Parrot.compute_cost, both parameter groups are clipped + stepped and checkpointed, and sample.py runs
    parrot.sampleRnn.sample_raw on the generated frames."""
    monkeypatch.setenv("RESULTS_DIR", str(tmp_path))
    sys.path.insert(0, ROOT)
    import importlib
    from parrot_amd.checkpoint import load_parameters
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    train = importlib.import_module("train")
    sample = importlib.import_module("sample")
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=32, EMB_SIZE=8)
    try:
        argv = ["--experiment_name", "raw", "--rnn_h_dim", "64", "--readouts_dim", "64", "--batch_size", "2",
                "--seq_size", "20", "--num_layers", "1", "--labels_type", "text", "--synthetic_examples", "4",
                "--save_every", "3", "--max_steps", "6", "--save_dir", str(tmp_path), "--raw_output", "1",
                "--lr_schedule", "1"]
        train.main(argv)
        best = os.path.join(str(tmp_path), "vctk", "pkl", "best_raw.tar")
        vals = load_parameters(best)
        assert any(k.startswith('/parrot/samplernn/SampleLevel.') for k in vals), "SampleRNN group checkpointed"
        assert '/parrot/rnn1.state_to_gates' in vals
        gen_x, lengths = sample.main(["--experiment_name", "raw", "--num_samples", "2", "--num_steps", "12",
                                      "--save_dir", str(tmp_path)])
        assert gen_x.shape == (2, 12, 63)
        raw_dir = os.path.join(str(tmp_path), "vctk", "samples", "new_raw")
        assert len([f for f in os.listdir(raw_dir) if f.endswith('.wav')]) == 2
    finally:
        lib.delete_all_params()
        tt.configure(DIM=1024, EMB_SIZE=256)


def test_pinned_loader_reuses_a_bounded_ring(dev):
    from parrot_amd.datasets import PinnedAsyncLoader, parrot_stream
    stream = parrot_stream('vctk', batch_size=4, seq_size=20, labels_type='text', raw_data=False, num_examples=16,
                           sorting_mult=1)
    loader = PinnedAsyncLoader(stream, dev, depth=2)
    n = 0
    for epoch in range(3):
        for b in loader:
            assert b['features'].is_cuda and torch.isfinite(b['features']).all()
            n += 1
    assert n >= 12
    # 4 slots x 4 array sources: independent of how many batches went through
    assert loader.pinned_allocations <= 4 * 4 * 2, loader.pinned_allocations
