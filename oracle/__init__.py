"""CPU oracle (test infrastructure only; see master_oracle.py). Never imported by the product package."""
