"""Throw-away stand-in for the Biopython package, used ONLY by tools/refharness/make_nr_goldens.py to let the REAL
/root/reference/scripts/nr_flt.py run in the build container (Biopython is absent and cannot be installed).  Test
infrastructure: never imported by the product."""
