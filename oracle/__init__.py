"""CPU oracle for the RADIAN hot path -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py)."""
