# Blocks mode: multi-scale 224x224 tiles per image (reference configs/oake/blocks.py).
_base_ = ['base.py']
train, val = (dict(dataloader=dict(dataset=dict(output_dir=f'data/coco/oake/blocks/{s}2017')))
              for s in ('train', 'val'))
log = dict(interval=10)
