# Objects mode on LVIS v1 (COCO images, LVIS annotations; reference configs/oake/objects_lvis.py).
_base_ = ['objects_coco.py']


def _split(name):
    return dict(dataloader=dict(dataset=dict(
        type='LVISDataset', root='data/coco',
        annFile=f'data/lvis_v1/annotations/lvis_v1_{name}.json',
        output_dir=f'data/lvis_v1/oake/objects/{name}2017',
        proposal_file=f'data/lvis_v1/proposals/oln_r50_fpn_lvis_{name}.pkl')))


train, val = _split('train'), _split('val')
