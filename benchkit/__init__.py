"""Bench / test infrastructure that is NOT part of the product package: tokenizer and processor stand-ins for runs without vocabulary
files (bench.py, scripts/, tests/). spider_amd/ holds only what the hot path needs."""
