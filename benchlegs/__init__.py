"""bench.py's legs, rig and routes (see bench.py)."""
