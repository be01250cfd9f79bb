# Dataset roots shared by the three OAKE modes (same keys as the reference's configs/oake/base.py).
_COCO = 'data/coco'
train = dict(
    dataloader=dict(
        dataset=dict(root=f'{_COCO}/train2017',
                     annFile=f'{_COCO}/annotations/instances_train2017.json'),
        num_workers=2,
    ),
)
val = dict(
    dataloader=dict(
        dataset=dict(root=f'{_COCO}/val2017',
                     annFile=f'{_COCO}/annotations/instances_val2017.json'),
        num_workers=2,
    ),
)
# crops per encoder pass (build-side batching; the reference encodes one image at a time)
batch_size = 256

# MI355X fast path (not in the reference): decode + preprocess on the device —
#   --override .train.dataloader.dataset.device_decode:True .val.dataloader.dataset.device_decode:True
# (the validator then runs without DataLoader workers: files are read by a prefetch thread of the process,
# Huffman passes run on `decode_threads` native threads; measured per GPU, files -> .pth: 5-6 k images/s for
# globals (12 k with two processes per GPU), 85-94 k crops/s for blocks, 23.6 k crops/s for objects —
# docs/history/design_sections_5_6_as_of_round5.md §5.5, profiles/r02_sweep_1gpu.log)
decode_threads = 32

# Behaviours of the un-vendored LutingWang/CLIP fork / todd that the reference's sources do not pin
# (oadp_amd/clip/settings.py).  The values below are our reading of the
# call sites; flip them here (or with --override .fork.load_default_true:center_crop ...) if the fork
# says otherwise — no code change needed.
fork = dict(
    load_default_true='squash',                                            # clip.load_default(True) transform
    positional_interpolation=dict(mode='bicubic', align_corners=False),    # visual.interpolate_positional_embedding
    min_wh_inclusive=True,                                                 # todd BBoxes.indices(min_wh): >= (True) or >
)
