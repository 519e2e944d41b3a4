"""CPU oracle for the hot path -- test infrastructure only (see oracle/torch_ref.py, oracle/oracle.c)."""
