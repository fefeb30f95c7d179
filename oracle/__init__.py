"""TEST INFRASTRUCTURE -- CPU oracle for the self-consistency aggregation path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product package ``o1_inference_scaling_laws_amd`` never does.

* ``oracle.pyoracle``  -- line-by-line Python restatement of /root/reference/o1.py:181-247 that calls
  ``statistics.multimode`` exactly like the reference (small cases, arbitrary Python ints).
* ``oracle.coracle``   -- ctypes binding of ``scv_oracle.c`` (same integer outputs as include/scvote.h;
  fast enough for N = 2^20 cells).
* ``oracle.ref_harness`` -- runs the UNMODIFIED reference under stub modules (build container only;
  /root/reference does not exist on the GPU box) to generate tests/golden/.
"""
