# Globals mode: one CLIP-preprocessed crop per image (reference configs/oake/globals.py).
_base_ = ['base.py']
train, val = (dict(dataloader=dict(dataset=dict(output_dir=f'data/coco/oake/globals/{s}2017')))
              for s in ('train', 'val'))
log = dict(interval=50)
