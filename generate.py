"""Kept entry point of reference generate.py (vocoder synthesis); see parrot_amd/generate.py."""
from parrot_amd.generate import generate_wav  # noqa: F401
