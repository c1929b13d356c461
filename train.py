"""Training entry point -- mirrors reference train.py: parse args, dump the config, build the streams,
build Parrot, then StepClipping(10*grad_clip) o Adam(lr) steps with TBPTT carry, checkpoints of the
best / last parameters.  Blocks' MainLoop / monitoring extensions are replaced by a plain loop;
with torchrun (WORLD_SIZE > 1) every rank trains on its shard of each batch and gradients are
all-reduced with RCCL (parrot_amd/dist.py)."""
import os
import pickle
import time

import numpy
import torch

from parrot_amd import dist as pdist
from parrot_amd.bricks import Constant, IsotropicGaussian
from parrot_amd.checkpoint import dump_parameters, load_parameters
from parrot_amd.datasets import PinnedAsyncLoader, parrot_stream
from parrot_amd.model import Parrot
from parrot_amd.trainer import Trainer
from parrot_amd.utils import train_parse


def save_parameters(path, parrot, extra=None):
    """Blocks-style checkpoint (tar with a `_parameters` npz, train.py:157-173) + the TBPTT carry."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    carry = {}
    for B, c in parrot._carry.items():
        if not isinstance(B, int):
            continue
        for l, h in enumerate(c['h']):
            carry['B%d|last_h%d' % (B, l + 1)] = h.detach().cpu().numpy()
        carry['B%d|last_k' % B] = c['k'].detach().cpu().numpy()
        carry['B%d|last_w' % B] = c['w'].detach().cpu().numpy()
    dump_parameters(path, parrot.get_parameter_values(), carry)


def main(argv=None):
    args = train_parse(argv)
    rank, local_rank, world = pdist.init_process_group()
    exp_name, save_dir = args.experiment_name, args.save_dir
    if rank == 0:
        os.makedirs(os.path.join(save_dir, 'config'), exist_ok=True)
        with open(os.path.join(save_dir, 'config', exp_name + '.pkl'), 'wb') as f:  # train.py:25-28
            pickle.dump(args, f, protocol=2)

    labels_type = args.labels_type if args.labels_type in ('text', 'unaligned_phonemes') else 'text'
    assert args.batch_size % world == 0, "global batch must divide over the ranks"
    local_batch = args.batch_size // world

    def stream(which):  # every rank draws the same global batches and keeps its contiguous shard
        return parrot_stream(args.dataset, args.use_speaker, (which,), args.batch_size,
                             noise_level=args.feedback_noise_level, labels_type=labels_type,
                             seq_size=args.seq_size, raw_data=False, num_examples=args.synthetic_examples)

    train_stream, valid_stream = stream('train'), stream('valid')
    device = torch.device(args.device, local_rank) if args.device == 'cuda' else torch.device(args.device)
    parrot = Parrot(
        input_dim=args.input_dim, output_dim=args.output_dim, rnn_h_dim=args.rnn_h_dim,
        readouts_dim=args.readouts_dim, weak_feedback=args.weak_feedback, full_feedback=args.full_feedback,
        feedback_noise_level=args.feedback_noise_level, layer_norm=args.layer_norm,
        use_speaker=args.use_speaker, num_speakers=args.num_speakers, speaker_dim=args.speaker_dim,
        which_cost=args.which_cost, num_characters=args.num_characters, attention_type=args.attention_type,
        attention_alignment=args.attention_alignment, encoder_type=args.encoder_type,
        weights_init=IsotropicGaussian(0.01), biases_init=Constant(0.), raw_output=False, name='parrot',
        num_layers=args.num_layers, encoder_literal=bool(args.encoder_literal), device=device,
        use_graph=bool(args.use_graph))
    parrot.initialize()
    if args.load_experiment:
        parrot.set_parameter_values(load_parameters(
            os.path.join(save_dir, 'pkl', 'best_' + args.load_experiment + '.tar')))
    trainer = Trainer(parrot, learning_rate=args.learning_rate, grad_clip=args.grad_clip)
    lo, hi = pdist.shard_batch(args.batch_size, rank, world)

    def shard(batch):
        out = dict(batch)
        for k in ('features', 'features_mask'):
            out[k] = batch[k][:, lo:hi].contiguous()
        for k in ('labels', 'labels_mask', 'speaker_index'):
            if k in batch:
                out[k] = batch[k][lo:hi].contiguous()
        return out

    def evaluate():
        tot, n = 0.0, 0
        with torch.no_grad():
            for b in PinnedAsyncLoader(valid_stream, device):
                b = shard(b)
                c, upd, _, _ = parrot.compute_cost(b['features'], b['features_mask'], b['labels'],
                                                   b['labels_mask'], b.get('speaker_index'), b['start_flag'],
                                                   local_batch)
                parrot.apply_updates(upd)
                tot, n = tot + float(c), n + 1
        return tot / max(n, 1)

    best, it, t0 = float('inf'), 0, time.time()
    done = False
    while not done:
        for b in PinnedAsyncLoader(train_stream, device):
            b = shard(b)
            cost = trainer.step(b['features'], b['features_mask'], b['labels'], b['labels_mask'],
                                b.get('speaker_index'), b['start_flag'])
            it += 1
            if it % args.save_every == 0 or (args.max_steps and it >= args.max_steps):
                valid = evaluate()
                if rank == 0:
                    print("iter %d train_%s %.5f valid_%s %.5f (%.1f s)" % (
                        it, args.which_cost, float(cost), args.which_cost, valid, time.time() - t0), flush=True)
                    save_parameters(os.path.join(save_dir, 'pkl', 'last_' + exp_name + '.tar'), parrot,
                                    dict(iterations=it))
                    if valid < best:
                        best = valid
                        save_parameters(os.path.join(save_dir, 'pkl', 'best_' + exp_name + '.tar'), parrot,
                                        dict(iterations=it, valid=valid))
                if not numpy.isfinite(valid) and args.lr_schedule:
                    trainer.cut_learning_rate()
            if (args.max_steps and it >= args.max_steps) or \
                    (args.time_limit and time.time() - t0 > 3600. * args.time_limit):
                done = True
                break
    if rank == 0:
        print("Training finished after %d iterations." % it)
    parrot.close()


if __name__ == "__main__":
    main()
