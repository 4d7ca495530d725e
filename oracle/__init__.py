"""CPU oracle for the SLFP conv2d path -- test infrastructure only (see oracle/README.md)."""
