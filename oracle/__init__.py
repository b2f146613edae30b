"""CPU oracle (test infrastructure only -- see tpgsr_oracle.py header)."""
