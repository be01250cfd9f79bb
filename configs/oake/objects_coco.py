_base_ = ['base.py']
_OUT = 'data/coco/oake/objects'
_PROP = 'data/coco/proposals'
train = dict(dataloader=dict(dataset=dict(
    type='COCODataset', output_dir=f'{_OUT}/train2017',
    proposal_file=f'{_PROP}/oln_r50_fpn_coco_train.pkl', proposal_sorted=True)))
val = dict(dataloader=dict(dataset=dict(
    type='COCODataset', output_dir=f'{_OUT}/val2017',
    proposal_file=f'{_PROP}/oln_r50_fpn_coco_val.pkl', proposal_sorted=True)))
log = dict(interval=5)
mini_batch_size = 512
batch_size = 512
