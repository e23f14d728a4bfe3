"""Stand-in for the absent, unpinned pip dependency e3nn (requirements.txt:21), used ONLY when
oracle/ref_import.py runs the reference's GaussianAdapter in the build container: the two
functions the reference calls (sh_rotation.py:4) are oracle/adapter_ref.py's restatement of
e3nn's published algorithm.  Everything else of the reference runs unmodified."""
