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
