"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see pfm_oracle.c).  Never imported by rnascan_amd/."""
