"""OAKE feature extraction (``oadp.oake.globals / blocks / objects``) on the MI355X encoder.

Mirrors the reference's module / class / method names for this path so the
parity tests read like the reference: ``Dataset._partition``, ``_partitions``, ``_bbox``,
``_expand``, ``_mask``, ``_preprocess``; ``Validator._build_model``, ``_run_iter``, ``main``.
The per-image ``<output_dir>/<image_id:012d>.pth`` contract is unchanged; what
changes is that crops from many images are encoded together (the reference runs batch 1)."""
