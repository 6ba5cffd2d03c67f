"""CPU oracle for the optik random-restart IK hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``optik_amd``) never does.

``oracle.binding``    ctypes view of ``liboptik_oracle.so`` (C restatement).
``oracle.urdf_chain`` Python restatement of the reference's URDF -> chain loader
                      (/root/reference/crates/optik/src/kinematics.rs:18-105, 263-319).
"""
