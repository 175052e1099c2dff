"""CPU oracle for the neural-audio-codec forward path.

TEST INFRASTRUCTURE ONLY.  Nothing in here is product code: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The shipped path is the HIP library in
``audio_generation_amd/csrc`` and it fails loudly when that library is missing.

What it restates (all citations relative to the upstream reference tree):

* ``oracle.codec``      -- causal conv stacks + VQAE wiring, ``networks/vae.py:14-322``
                           and the weight-norm helper ``networks/utils.py:34-42``.
* ``oracle.rvq``        -- residual vector quantiser.  The upstream source
                           (module ``som_quantizer`` from the author's un-pinned
                           "quantization-maps" repository, imported at
                           ``networks/vae.py:6``) is NOT in the reference tree, so
                           this is a restatement of the published algorithm
                           (README.md:48) anchored on the call sites
                           ``vae.py:245-251,315-318,333``.  **Parity unpinned.**
* ``oracle.attention``  -- ALiBi + pre-LN attention + FFN, ``networks/transformers.py:7-279``.
* ``oracle.wavelets``   -- multires cascade + wavelet layer, ``networks/wavelets.py:8-234``.

Pinning: ``tests/test_oracle_golden.py`` checks every function against the
fixtures in ``tests/golden/`` which were produced by running the reference's
own modules in the build container (``tests/golden/make_goldens.py``).
"""
