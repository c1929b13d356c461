"""Training entry point -- mirrors reference train.py: parse args, dump the config, build the streams,
build Parrot, then StepClipping(10*grad_clip) o Adam(lr) steps with TBPTT carry, validation every `save_every`
iterations, checkpoints of the best / last parameters, optional LearningRateSchedule (patience 10, 5 cuts,
train.py:175-182).  Blocks' MainLoop / monitoring extensions are replaced by a plain loop; with torchrun
(WORLD_SIZE > 1) every rank trains on its shard of each batch (the producer materialises only that shard) and
gradients are all-reduced with RCCL (parrot_amd/dist.py).  Every decision that changes control flow or optimiser
state (stop, learning-rate cut, best checkpoint) is taken on collectively reduced values, so the ranks never
diverge."""
import os
import pickle
import time

import torch

from parrot_amd import dist as pdist
from parrot_amd.bricks import Constant, IsotropicGaussian
from parrot_amd.checkpoint import dump_parameters, load_parameters
from parrot_amd.datasets import PinnedAsyncLoader, parrot_stream
from parrot_amd.model import Parrot
from parrot_amd.trainer import LearningRateSchedule, Trainer
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
    values = dict(parrot.get_parameter_values())
    if getattr(parrot, 'raw_output', False):  # SampleRNN head: registry names under a /sampleRnn/ prefix
        from parrot_amd.sampleRNN import lib as srn_lib
        for n, t in srn_lib.named_params().items():
            values['/parrot/samplernn/' + n] = t.detach().cpu().numpy()
    dump_parameters(path, values, carry)


def restore_parameters(path, parrot):
    values = load_parameters(path)
    srn = {k[len('/parrot/samplernn/'):]: v for k, v in values.items() if k.startswith('/parrot/samplernn/')}
    parrot.set_parameter_values({k: v for k, v in values.items() if not k.startswith('/parrot/samplernn/')})
    if srn:
        from parrot_amd.sampleRNN import lib as srn_lib
        srn_lib.set_params(srn)


def main(argv=None):
    args = train_parse(argv)
    rank, local_rank, world = pdist.init_process_group()
    exp_name, save_dir = args.experiment_name, args.save_dir
    if rank == 0:
        os.makedirs(os.path.join(save_dir, 'config'), exist_ok=True)
        with open(os.path.join(save_dir, 'config', exp_name + '.pkl'), 'wb') as f:  # train.py:25-28
            pickle.dump(args, f, protocol=2)

    if args.labels_type not in ('text', 'unaligned_phonemes'):
        raise SystemExit(
            "--labels_type %r: only 'text' and 'unaligned_phonemes' fit Parrot's integer label matrix "
            "(model.py:511; the reference's own default 'full_labels' cannot run there either)" % args.labels_type)
    assert args.batch_size % world == 0, "global batch must divide over the ranks"
    local_batch = args.batch_size // world
    raw_output = bool(args.raw_output)

    def stream(which, noise_level):
        # every rank forms the same global batches and materialises only its contiguous shard
        return parrot_stream(args.dataset, args.use_speaker, (which,), args.batch_size, noise_level=noise_level,
                             labels_type=args.labels_type, seq_size=args.seq_size, raw_data=raw_output,
                             num_examples=args.synthetic_examples, shard=(rank, world) if world > 1 else None)

    train_stream = stream('train', args.feedback_noise_level)
    # validation runs without feedback noise (train.py:40-48)
    valid_stream = stream('valid', None if args.feedback_noise_level is None else 0.)
    device = torch.device(args.device, local_rank) if args.device == 'cuda' else torch.device(args.device)
    parrot = Parrot(
        input_dim=args.input_dim, output_dim=args.output_dim, rnn_h_dim=args.rnn_h_dim,
        readouts_dim=args.readouts_dim, weak_feedback=args.weak_feedback, full_feedback=args.full_feedback,
        feedback_noise_level=args.feedback_noise_level, layer_norm=args.layer_norm,
        use_speaker=args.use_speaker, num_speakers=args.num_speakers, speaker_dim=args.speaker_dim,
        which_cost=args.which_cost, num_characters=args.num_characters, attention_type=args.attention_type,
        attention_alignment=args.attention_alignment, encoder_type=args.encoder_type,
        weights_init=IsotropicGaussian(0.01), biases_init=Constant(0.), raw_output=raw_output, name='parrot',
        num_layers=args.num_layers, cell_type=args.cell_type, compute_dtype=args.compute_dtype, encoder_literal=bool(args.encoder_literal), device=device,
        use_graph=bool(args.use_graph))
    parrot.initialize()
    best_path = os.path.join(save_dir, 'pkl', 'best_' + exp_name + '.tar')
    last_path = os.path.join(save_dir, 'pkl', 'last_' + exp_name + '.tar')
    if args.load_experiment:
        restore_parameters(os.path.join(save_dir, 'pkl', 'best_' + args.load_experiment + '.tar'), parrot)
    trainer = Trainer(parrot, learning_rate=args.learning_rate, grad_clip=args.grad_clip)

    def reload_best():  # LearningRateSchedule: model.set_parameter_values(load_parameters(best)) on every rank
        pdist.barrier()  # rank 0 has finished writing the file
        if os.path.exists(best_path):
            restore_parameters(best_path, parrot)
        for p_, _ in trainer.groups:
            pdist.broadcast_parameters_(p_)

    schedule = LearningRateSchedule(trainer, reload_best, patience=10, num_cuts=5) if args.lr_schedule else None

    def noise_for(b):
        lvl = b.get('feedback_noise_level')
        if lvl is None or not parrot.weak_feedback:
            return None
        return float(lvl) * torch.randn_like(b['features'][:-1]) if float(lvl) > 0 else \
            torch.zeros_like(b['features'][:-1])

    def evaluate():
        """DataStreamMonitoring on the validation stream (train.py:116-121): the masked-mean cost over the whole
        (global) stream.  It never applies `extra_updates` to the TRAINING carry: the validation windows chain
        through a carry of their own, and the training carry is put back afterwards."""
        saved = parrot._carry
        parrot._carry = {}
        num = den = 0.0
        try:
            with torch.no_grad():
                for b in PinnedAsyncLoader(valid_stream, device):
                    c, upd, _, _ = parrot.compute_cost(b['features'], b['features_mask'], b['labels'],
                                                       b['labels_mask'], b.get('speaker_index'), b['start_flag'],
                                                       local_batch, raw_audio=b.get('raw_audio'),
                                                       feedback_noise=noise_for(b))
                    parrot.apply_updates(upd)
                    d_ = float(b['features_mask'][1:].sum())
                    num, den = num + float(c) * (d_ + pdist.COST_EPS), den + d_
        finally:
            parrot._carry = saved
        num, den = pdist.sum_over_ranks([num, den])  # the same value on every rank
        return num / (den + pdist.COST_EPS)

    best, it, t0 = float('inf'), 0, time.time()
    done = False
    while not done:
        for b in PinnedAsyncLoader(train_stream, device):
            cost = trainer.step(b['features'], b['features_mask'], b['labels'], b['labels_mask'],
                                b.get('speaker_index'), b['start_flag'], feedback_noise=noise_for(b),
                                raw_audio=b.get('raw_audio'))
            it += 1
            finish = False
            if it % args.save_every == 0 or (args.max_steps and it >= args.max_steps):
                valid = evaluate()
                if rank == 0:
                    print("iter %d train_%s %.5f valid_%s %.5f lr %.3g (%.1f s)" % (
                        it, args.which_cost, float(cost), args.which_cost, valid, trainer.lr, time.time() - t0),
                        flush=True)
                    save_parameters(last_path, parrot, dict(iterations=it))
                    if valid < best:  # TrackTheBest + Checkpoint(best_*) (train.py:138-160)
                        save_parameters(best_path, parrot, dict(iterations=it, valid=valid))
                best = min(best, valid) if valid == valid else best
                if schedule is not None:
                    cut, finish = schedule.update(valid)  # `valid` is already global: same decision on every rank
                    if cut and rank == 0:
                        print("learning rate cut to %.3g (cut %d of %d), best parameters reloaded" % (
                            trainer.lr, schedule.count_cuts, schedule.num_cuts), flush=True)
            # `finish` (from the global validation cost) and the step limit are the same on every rank; wall clocks
            # are not, so the time limit is decided collectively (every 8th iteration: the agreement reads a flag
            # back to the host, which would otherwise drain the launch queue every step)
            stop = finish or bool(args.max_steps and it >= args.max_steps)
            if args.time_limit and it % 8 == 0:
                stop = pdist.any_rank(stop or time.time() - t0 > 3600. * args.time_limit)
            if stop:
                done = True
                break
    if rank == 0:
        print("Training finished after %d iterations." % it)
    parrot.close()


if __name__ == "__main__":
    main()
