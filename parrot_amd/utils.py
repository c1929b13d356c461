"""Argument parsers of the entry points -- same flags, defaults and quirks as reference utils.py
(train_parse utils.py:173-253, sample_parse utils.py:256-327), including the `type=bool` flags for
which any non-empty string is truthy.  Plotting / animation helpers (utils.py:30-170) are cosmetic
and out of scope.

Extra flags (not in the reference): --num_layers, --cell_type, --compute_dtype, --encoder_literal, --synthetic_examples, --max_steps,
--device, --use_graph.
"""
from __future__ import annotations

import argparse
import os

SPTK_DIR = '/data/lisatmp4/kumarkun/merlin/tools/bin/SPTK-3.9/'   # utils.py:20
WORLD_DIR = '/data/lisatmp4/kumarkun/merlin/tools/bin/WORLD/'     # utils.py:21


def _results_dir():
    return os.environ.get('RESULTS_DIR', os.path.join(os.getcwd(), 'results'))


def _common_extras(parser):
    parser.add_argument('--num_layers', type=int, default=3, help='decoder GRU layers (reference: 3)')
    parser.add_argument('--cell_type', type=str, default='gru', choices=('gru', 'lstm'),
                        help="decoder layers: 'gru' (the reference's GatedRecurrent) or 'lstm' (BASELINE configs[3])")
    parser.add_argument('--compute_dtype', type=str, default='float32', choices=('float32', 'bf16'),
                        help='bf16: bf16 MFMA operands, f32 accumulation / states / master weights (DESIGN.md 3.4)')
    parser.add_argument('--encoder_literal', type=int, default=1,
                        help='1: scan the encoder over the batch axis like the reference does')
    parser.add_argument('--device', type=str, default='cuda')
    parser.add_argument('--use_graph', type=int, default=1)
    parser.add_argument('--synthetic_examples', type=int, default=64)


def train_parse(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--experiment_name', type=str, default='baseline')
    parser.add_argument('--encoder_type', type=str, default='bidirectional')
    parser.add_argument('--encoder_dim', type=int, default=128)
    parser.add_argument('--input_dim', type=int, default=420)
    parser.add_argument('--output_dim', type=int, default=63)
    parser.add_argument('--rnn_h_dim', type=int, default=1024)
    parser.add_argument('--readouts_dim', type=int, default=1024)
    parser.add_argument('--weak_feedback', type=bool, default=False)
    parser.add_argument('--full_feedback', type=bool, default=False)
    parser.add_argument('--feedback_noise_level', type=float, default=None)
    parser.add_argument('--layer_norm', type=bool, default=False)
    parser.add_argument('--labels_type', type=str, default='text')  # reference default 'full_labels' cannot feed model.py:511 (imatrix)
    parser.add_argument('--which_cost', type=str, default='MSE')
    parser.add_argument('--attention_type', type=str, default='graves')
    parser.add_argument('--attention_alignment', type=float, default=1.)
    parser.add_argument('--num_characters', type=int, default=43)
    parser.add_argument('--batch_size', type=int, default=8)
    parser.add_argument('--seq_size', type=int, default=50)
    parser.add_argument('--save_every', type=int, default=500)
    parser.add_argument('--learning_rate', type=float, default=1e-4)
    parser.add_argument('--grad_clip', type=float, default=0.9)
    parser.add_argument('--lr_schedule', type=bool, default=False)
    parser.add_argument('--load_experiment', type=str, default=None)
    parser.add_argument('--raw_output', type=bool, default=False)
    parser.add_argument('--time_limit', type=float, default=None)
    parser.add_argument('--use_speaker', type=bool, default=False)
    parser.add_argument('--num_speakers', type=int, default=22)
    parser.add_argument('--speaker_dim', type=int, default=128)
    parser.add_argument('--dataset', type=str, default='vctk')
    parser.add_argument('--save_dir', type=str, default=_results_dir())
    parser.add_argument('--max_steps', type=int, default=None, help='stop after this many windows')
    _common_extras(parser)
    args = parser.parse_args(argv)
    if args.dataset not in args.save_dir:
        args.save_dir = os.path.join(args.save_dir, args.dataset)
    return args


def sample_parse(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--experiment_name', type=str, default='baseline')
    parser.add_argument('--sampling_bias', type=float, default=1.)
    parser.add_argument('--timing_coeff', type=float, default=1.)
    parser.add_argument('--sharpening_coeff', type=float, default=1.)
    parser.add_argument('--num_samples', type=int, default=10)
    parser.add_argument('--num_steps', type=int, default=2048)
    parser.add_argument('--samples_name', type=str, default='sample')
    parser.add_argument('--speaker_id', type=int, default=None)
    parser.add_argument('--mix', type=float, default=None)
    parser.add_argument('--dataset', type=str, default='vctk')
    parser.add_argument('--new_sentences', type=bool, default=False)
    parser.add_argument('--save_dir', type=str, default=_results_dir())
    parser.add_argument('--sptk_dir', type=str, default=SPTK_DIR)
    parser.add_argument('--world_dir', type=str, default=WORLD_DIR)
    parser.add_argument('--process_originals', type=bool, default=False)
    parser.add_argument('--do_post_filtering', type=bool, default=False)
    parser.add_argument('--animation', type=bool, default=False)
    parser.add_argument('--debug_plot', type=bool, default=False)
    parser.add_argument('--sample_one_step', type=bool, default=False)
    parser.add_argument('--use_last', type=bool, default=False)
    parser.add_argument('--phrase', type=str, default=None)
    parser.add_argument('--random_speaker', type=bool, default=False)
    parser.add_argument('--plot_raw', type=bool, default=False)
    parser.add_argument('--device', type=str, default='cuda')
    parser.add_argument('--synthetic_examples', type=int, default=64)
    args = parser.parse_args(argv)
    if args.dataset not in args.save_dir:
        args.save_dir = os.path.join(args.save_dir, args.dataset)
    return args


def end_of_utterance(phi, labels_length, num_steps, extra=40):
    """sample.py:145-163: first step whose window weight on the position just past the text exceeds the
    weight on every real character, plus `extra` frames; num_steps if never."""
    import numpy
    try:
        t = numpy.where((phi[:, labels_length, numpy.newaxis] > phi[:, :labels_length - 1]).all(axis=1))[0][0]
        return int(numpy.minimum(num_steps, t + extra))
    except Exception:
        return int(num_steps)
