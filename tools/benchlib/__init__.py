"""Parts of bench.py that are not the timed regions or the JSON line: the ctypes loader, the live counter passes, the clock sampler, the
multi-GPU legs and the rank launcher.  bench.py (the file the driver runs and hashes) imports them; the tests import them through it."""
