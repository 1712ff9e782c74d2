"""CPU oracle for the Contrastive-Lift rendering hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain CPU-PyTorch / numpy restatement of the
reference algorithm, written from the formulas, each function
citing the reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it -- and there only as the checker or as the
reported CPU baseline, never as the product path.  The product path (``contrastive_lift_amd``)
never imports ``oracle`` and fails loudly if the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * every function except ``render.dist_loss`` is pinned against outputs of the reference itself,
    generated in the build container by ``tests/golden/make_golden.py`` (which imports
    ``/root/reference`` with stubbed third-party modules) and committed as ``tests/golden/*.npz``;
  * ``render.dist_loss`` restates the public formula of ``torch_efficient_distloss==0.1.3``
    (reference ``requirements.txt:35``, call site
    ``model/renderer/panopli_tensoRF_renderer.py:101``).  The package is absent from the reference
    tree and from this image => **parity unpinned** for that one term.
"""
