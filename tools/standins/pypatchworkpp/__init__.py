"""Import-only stand-in (tools/gen_golden.py); never called on the hot path."""
