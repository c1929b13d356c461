"""Synthesis entry point -- mirrors reference sample.py: load the saved config + parameters, take one
test batch, run Parrot.sample_model (autoregressive decode, one hipGraph) or sample_using_input,
apply the end-of-utterance heuristic (sample.py:145-163) and hand the frames to generate_wav."""
import os
import pickle

import numpy
import torch

from parrot_amd.checkpoint import load_parameters
from parrot_amd.datasets import parrot_stream
from parrot_amd.generate import generate_wav
from parrot_amd.model import Parrot
from parrot_amd.utils import end_of_utterance, sample_parse


def main(argv=None):
    args = sample_parse(argv)
    with open(os.path.join(args.save_dir, 'config', args.experiment_name + '.pkl'), 'rb') as f:
        saved_args = pickle.load(f)
    assert saved_args.dataset == args.dataset
    params_mode = 'last_' if args.use_last else 'best_'
    args.samples_name = params_mode + args.samples_name
    parameters = load_parameters(os.path.join(args.save_dir, 'pkl', params_mode + args.experiment_name + '.tar'))
    srn_parameters = {k[len('/parrot/samplernn/'):]: v for k, v in parameters.items()
                      if k.startswith('/parrot/samplernn/')}
    parameters = {k: v for k, v in parameters.items() if not k.startswith('/parrot/samplernn/')}

    labels_type = saved_args.labels_type
    if labels_type not in ('text', 'unaligned_phonemes'):
        raise SystemExit("saved config has labels_type %r: only 'text' / 'unaligned_phonemes' fit Parrot's integer "
                         "label matrix (model.py:511)" % labels_type)
    raw_output = bool(getattr(saved_args, 'raw_output', False))  # sample.py:92-93
    test_stream = parrot_stream(args.dataset, saved_args.use_speaker, ('test',), args.num_samples,
                                args.num_steps, sorting_mult=1, labels_type=labels_type, raw_data=False,
                                num_examples=max(args.synthetic_examples, args.num_samples))
    data_tr = next(test_stream.get_epoch_iterator(as_dict=True))
    features_mask_tr = data_tr.get('features_mask')
    speaker_tr = data_tr.get('speaker_index')
    labels_tr, labels_mask_tr = data_tr.get('labels'), data_tr.get('labels_mask')
    if args.random_speaker:
        numpy.random.seed(1)
        speaker_tr = numpy.random.randint(1, saved_args.num_speakers, (args.num_samples, 1)).astype('int32')
    if args.speaker_id and saved_args.use_speaker:
        speaker_tr = speaker_tr * 0 + args.speaker_id

    device = torch.device(args.device)
    parrot = Parrot(
        input_dim=saved_args.input_dim, output_dim=saved_args.output_dim, rnn_h_dim=saved_args.rnn_h_dim,
        readouts_dim=saved_args.readouts_dim, weak_feedback=saved_args.weak_feedback,
        full_feedback=saved_args.full_feedback, feedback_noise_level=None, layer_norm=saved_args.layer_norm,
        use_speaker=saved_args.use_speaker, num_speakers=saved_args.num_speakers,
        speaker_dim=saved_args.speaker_dim, which_cost=saved_args.which_cost,
        num_characters=saved_args.num_characters, attention_type=saved_args.attention_type,
        attention_alignment=saved_args.attention_alignment, sampling_bias=args.sampling_bias,
        sharpening_coeff=args.sharpening_coeff, timing_coeff=args.timing_coeff,
        encoder_type=saved_args.encoder_type, raw_output=raw_output, name='parrot',
        num_layers=getattr(saved_args, 'num_layers', 3), cell_type=getattr(saved_args, 'cell_type', 'gru'),
        encoder_literal=bool(getattr(saved_args, 'encoder_literal', 1)),
        compute_dtype=getattr(saved_args, 'compute_dtype', 'float32'), device=device)
    print("Operand precision of the decoder: %s" % getattr(saved_args, 'compute_dtype', 'float32'))
    parrot.allocate()
    parrot.set_parameter_values(parameters)
    if raw_output:
        if not srn_parameters:
            raise ValueError("the experiment was trained with --raw_output but its checkpoint holds no "
                             "'/parrot/samplernn/*' parameters: the SampleRNN head would sample from its random "
                             "initialisation (noise)")
        from parrot_amd.sampleRNN import lib as srn_lib
        srn_lib.set_params(srn_parameters)
    print("Successfully loaded the parameters.")

    if args.sample_one_step:
        gen_x, gen_k, gen_w, gen_pi, gen_phi, gen_pi_att = parrot.sample_using_input(data_tr, args.num_samples)
    else:
        gen_x, gen_k, gen_w, gen_pi, gen_phi, gen_pi_att = parrot.sample_model(
            labels_tr, labels_mask_tr, features_mask_tr, speaker_tr, args.num_samples, args.num_steps)
    print("Successfully sampled the parrot.")

    gen_x, gen_phi = gen_x.swapaxes(0, 1), gen_phi.swapaxes(0, 1)
    features_lengths = []
    for idx in range(args.num_samples):
        ll = int(labels_mask_tr[idx].sum())
        features_lengths.append(end_of_utterance(gen_phi[idx], min(ll, gen_phi.shape[2] - 1), args.num_steps))
    if raw_output:  # sample.py:165-173: SampleRNN vocoder on the generated frames
        print("Sampling and saving raw audio...")
        to_save_path = os.path.join(args.save_dir, 'samples', 'new_raw')
        os.makedirs(to_save_path, exist_ok=True)
        parrot.sampleRnn.sample_raw(gen_x.swapaxes(0, 1).copy(), features_lengths, args.samples_name, to_save_path)
        print("Successfully sampled raw audio...")
    out_dir = os.path.join(args.save_dir, 'samples')
    for idx, this_sample in enumerate(gen_x):
        generate_wav(this_sample[:features_lengths[idx]], out_dir, args.samples_name + '_' + str(idx),
                     sptk_dir=args.sptk_dir, world_dir=args.world_dir, norm_info_file=None,
                     do_post_filtering=args.do_post_filtering)
    print("Saved %d feature files under %s" % (args.num_samples, out_dir))
    parrot.close()
    return gen_x, features_lengths


if __name__ == "__main__":
    main()
