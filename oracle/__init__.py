"""CPU oracle for the latent score-model sampling hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy restatement of the
reference algorithm (DanielMitiku/score_based_multimodal_autoencoder) for the
path BASELINE.json's north_star names.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and there only as the
checker / reported CPU baseline -- never as the thing shipped.  The product
package (``score_based_multimodal_autoencoder_amd``) never imports ``oracle``.

Parity pin: every function here is checked against outputs of the real reference
modules, generated in the build container by ``tests/golden/make_golden.py`` and
committed as ``tests/golden/*.npz`` (the reference has no tests or golden vectors
of its own -- SURVEY.md section 4).

File:line citations are relative to the reference tree root.
"""
