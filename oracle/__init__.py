"""Test infrastructure only — see vispec_oracle.py."""
