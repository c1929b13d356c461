"""NumPy restatement of reference quantize.py (test infrastructure, see oracle/__init__.py).

Follows quantize.py line by line; float64 arithmetic, same operation order, so integer outputs are
bit-identical to the reference on the same NumPy build.
"""
import numpy as np


def normalize(data):
    """quantize.py:14-18 -- per-row shift to min 0 then divide by the row max (in place)."""
    data -= data.min(axis=1)[:, None]
    data /= data.max(axis=1)[:, None]
    return data


def linear_quantize(data, q_levels):
    """quantize.py:20-36."""
    eps = np.float64(1e-5)
    data *= (q_levels - eps)
    data += eps / 2
    return data.astype('int32')


def linear2mu(x, mu=255):
    """quantize.py:44-66."""
    x_mu = np.sign(x) * np.log(1 + mu * np.abs(x)) / np.log(1 + mu)
    return ((x_mu + 1) / 2 * mu).astype('int16')


def mu2linear(x, mu=255):
    """quantize.py:68-78."""
    mu = float(mu)
    x = x.astype('float32')
    y = 2. * (x - (mu + 1.) / 2.) / (mu + 1.)
    return np.sign(y) * (1. / mu) * ((1. + mu) ** np.abs(y) - 1.)


def batch_quantize(data, q_levels, q_type):
    """quantize.py:83-99 (`__batch_quantize`)."""
    data = data.astype('float64')
    data = normalize(data)
    if q_type == 'linear':
        return linear_quantize(data, q_levels)
    if q_type == 'mu-law':
        data = 2. * data - 1.
        return linear2mu(data)
    raise NotImplementedError(q_type)
