"""CPU oracle (test infrastructure only; see xggm_oracle.py header)."""
