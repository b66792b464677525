"""CPU oracle for the D-VAE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``d-vae_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  ``dense_ref`` is checked against the real reference
(``/root/reference/dvae/models/dvae_model.py`` imported with two arithmetic-free
stubs) by ``tests/golden/make_golden.py``; its outputs are committed as
``tests/golden/*.npz`` and re-checked by ``tests/test_oracle_golden.py``.
"""
