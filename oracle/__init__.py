"""CPU oracle for the VideoSeal embed -> (augment) -> extract hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``videoseal_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / reported baseline.

It is a functional fp32 restatement (torch CPU ops) of the reference algorithm,
each function citing the reference file:line it follows.  It is pinned against
the real reference: ``tests/golden/make_golden.py`` imports the unmodified
modules from ``/root/reference`` (stub-import recipe of SURVEY.md appendix A),
loads the same seeded state_dict and stores the outputs under ``tests/golden``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors.

Third-party arithmetic that is not in /root/reference:
  * torch ATen conv2d / interpolate / layer_norm / embedding -- installed here,
    so ATen *is* the oracle for those primitives.
  * torchvision.transforms.functional colour/blur ops -- NOT installed and NOT
    vendored by the reference (unpinned in its pyproject).  ``oracle/augment.py``
    restates the published torchvision ``_functional_tensor`` semantics;
    parity for those ops is "unpinned" (no reference fixture can be generated).
  * Pillow/libjpeg-turbo (JPEG) -- installed; the PIL round trip is the oracle.
"""
