_base_ = ['objects_coco.py']
_OUT = 'data/lvis_v1/oake/objects'
_PROP = 'data/lvis_v1/proposals'
train = dict(dataloader=dict(dataset=dict(
    type='LVISDataset', root='data/coco', annFile='data/lvis_v1/annotations/lvis_v1_train.json',
    output_dir=f'{_OUT}/train2017', proposal_file=f'{_PROP}/oln_r50_fpn_lvis_train.pkl')))
val = dict(dataloader=dict(dataset=dict(
    type='LVISDataset', root='data/coco', annFile='data/lvis_v1/annotations/lvis_v1_val.json',
    output_dir=f'{_OUT}/val2017', proposal_file=f'{_PROP}/oln_r50_fpn_lvis_val.pkl')))
