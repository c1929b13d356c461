"""Run the reference's own Python sources here (TEST INFRASTRUCTURE, used only by tests/golden/make_ref_golden.py and
tests/test_oracle_cpu.py): eager torch-backed stand-ins for Theano and Blocks (theano_shim.py, blocks_shim.py) plus a
loader that executes /root/reference files unmodified under Python-2 semantics (loader.py)."""
