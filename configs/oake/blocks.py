_base_ = ['base.py']
_OUT = 'data/coco/oake/blocks'
train = dict(dataloader=dict(dataset=dict(output_dir=f'{_OUT}/train2017')))
val = dict(dataloader=dict(dataset=dict(output_dir=f'{_OUT}/val2017')))
log = dict(interval=10)
