"""CPU oracle for the LINR-PCGC hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``linr_pcgc_amd/``) never imports it and has no CPU fallback.

Contents
  octree.py       numpy restatement of the per-frame multi-scale prep (custom_dataset.py:259-355)
  network.py      torch-CPU fp32/fp64 restatement of the coding network (model_core.py, upsample.py,
                  resnet.py) in dense-row form over a neighbour table
  torchac_port.c  plain-C restatement of torchac 0.9.3's range coder (third-party, un-vendored)
  ac.py           ctypes loader for torchac_port.c + the float->int16 CDF conversion
  model_codec.py  weight quantiser + Laplace CDF of model_compression/model_size_est.py

Pinning status (see DESIGN.md "Oracle"):
  * octree/occupancy/offset prep, bitstream packing: pinned by fixtures generated from the reference's
    own pure-torch helpers (tests/golden/make_golden.py).
  * weight quantiser: pinned by loot/gop_32_62/70/side_info.json (mu, b, min, max).
  * range coder + Laplace-CDF quirk: pinned by the model stream implied by loot/gop_32_62/70/result.json (282,642 bits):
    35,319 bytes with the CPU's pdf (+ a 90-bit header of an older revision), or 35,320 bytes under today's 82-bit header
    if CUDA's pdf differs in the last bit of one of the CDF entries that sit on a rounding boundary - both consistent,
    tests/test_oracle_golden.py::test_model_stream_known_answer computes both.
  * network arithmetic (MinkowskiEngine semantics): PARITY UNPINNED at bit level - MinkowskiEngine 0.5.4 is not in
    /root/reference and not installable here; the restatement follows the reference call sites and
    MinkowskiEngine's documented semantics.  Pinned behaviourally by the checkpoint the reference ships
    (loot/gop_32_62/model.pth -> tests/golden/loot_model_kat.npz): the reference-trained weights predict unseen
    surfaces (0.96 bits/point) only under the assumed tap order / direction / wiring; each of the 11 other tap
    conventions gives 6.6-10.8 (tests/test_oracle_golden.py::test_reference_checkpoint_pins_network_semantics).
"""
