"""CPU oracle - test infrastructure only (see l2s_oracle.py header)."""
