"""Legs and plumbing of bench.py (repo root): one module per leg so that the line assembler (line.py) and the rank launcher are testable on CPU."""
