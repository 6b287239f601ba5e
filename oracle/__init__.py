"""CPU oracle for the multi-view point-tracking forward path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped product path:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU baseline.  ``mvtracker_amd`` never
imports this package and raises if its HIP library is missing.
"""
