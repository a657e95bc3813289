"""Throughput harness with the JSONL fields of reference vispec/evaluation/gen_spec_answer_*.py and the speed-up formula of speed.py."""
