# Objects mode on COCO: proposals from OLN (keys and paths as the reference's configs/oake/objects_coco.py).
_base_ = ['base.py']


def _split(name):
    return dict(dataloader=dict(dataset=dict(
        type='COCODataset',
        output_dir=f'data/coco/oake/objects/{name}2017',
        proposal_file=f'data/coco/proposals/oln_r50_fpn_coco_{name}.pkl',
        proposal_sorted=True)))


train, val = _split('train'), _split('val')
log = dict(interval=5)
mini_batch_size = 512   # crops per model.visual(objects, masks) call, as the reference
# crops gathered across images before they go down to the GPU (build-side batching).  One flush = one crop job list,
# one upload of masks, one native encode call that the library cuts into equal passes of <= 128 crops, one download:
# 8 images' worth keeps the fixed costs of a flush under 1 % (files -> .pth on one MI355X, 2000 images: 83.0 images/s
# at 1024, 90.4 at 2400 = the encoder-only rate of bench.py --mode objects; profiles/r04/sweep_objects_lookahead_batch.log)
batch_size = 2400
