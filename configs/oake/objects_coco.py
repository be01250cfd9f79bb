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
batch_size = 512        # crops gathered across images before an encoder pass (build-side batching)
