"""CPU oracle (test infrastructure).  See physicl_oracle.py; never imported by physicl_amd."""
