"""Baseline bidders (mirror of adcraft/baselines/): the cache arithmetic runs in the HIP engine."""
