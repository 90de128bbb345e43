"""CPU oracle for the flow-matching hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``particle_fm_amd`` may import this
package: it is the checker that the HIP path is compared against in
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg, never the thing that is shipped or measured as the product.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference's own
hot-path modules (``/root/reference/particle_fm/models/components/{epic,
time_emb,losses}.py`` and ``models/flow_matching_module.py``) in the build
container and records their outputs in ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors.
The fixed-step midpoint integrator lives in torchdyn (unpinned in
``requirements.txt:25``, absent from ``/root/reference``); it is restated from
its published algorithm in ``oracle/fm_ref.py::midpoint_trajectory`` and the
"midpoint" fixtures are labelled *reference vector field + restated integrator*.
"""
