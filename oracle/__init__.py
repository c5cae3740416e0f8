"""CPU oracle for the DT + NN-fill path.  TEST INFRASTRUCTURE ONLY (see oracle.py)."""
