"""Decode post-processing that needs no GPU."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_end_of_utterance_matches_the_reference_loop():
    """tests/golden/eou_golden.npz: the lengths the reference's own loop (sample.py:145-163, executed from the reference
    script by make_eou_golden.py) assigned to six seeded attention histories -- windows that pass the end of the text,
    one that is too slow, one that never moves."""
    from parrot_amd.utils import end_of_utterance
    gold = np.load(os.path.join(HERE, "golden", "eou_golden.npz"))
    phi, mask, num_steps = gold['phi'], gold['labels_mask'], int(gold['num_steps'])
    got = [end_of_utterance(phi[n], int(mask[n].sum()), num_steps) for n in range(phi.shape[0])]
    assert got == gold['features_lengths'].tolist()
    assert [int(mask[n].sum()) for n in range(phi.shape[0])] == gold['labels_lengths'].tolist()
