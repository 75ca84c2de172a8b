"""CPU oracle for the seigen explicit velocity-stress DG hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy/scipy restatement of
the algorithm that Firedrake/PyOP2 generate from ``seigen/elastic.py`` (the
reference; paths below are relative to the reference checkout).  It is the
checker for the HIP path and the CPU baseline of ``bench.py``; nothing under
``seigen_amd/`` may import it.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` use it.

Pinning status: the Firedrake stack (firedrake, pyop2, ufl, FIAT, tsfc, coffee,
petsc4py, mpi4py - unpinned, ``setup.py:18``; July-2017 Zenodo snapshots named
in ``tests/tiling/README.md:1-13``) is not installable in the build container
and the reference stores no golden vector of its own output, so bit-level
parity with Firedrake is "parity unpinned".  What the oracle IS pinned against
(tests/test_oracle_*.py):
  * the analytic eigenmode solutions of ``tests/eigenmode/eigenmode_2d.py:30-47``
    and ``eigenmode_3d.py:30-51`` (convergence at the expected DG order);
  * the receiver traces ``tests/explosive_source/REF-C1..3`` (external code,
    compared by ``uy.py`` in the reference);
  * polynomial-reproduction and energy-conservation properties of the weak
    form ``seigen/elastic.py:204-219``.
"""
