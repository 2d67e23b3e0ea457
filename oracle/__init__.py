"""CPU oracle (test infrastructure only): see oracle/vgpa_oracle.py."""
