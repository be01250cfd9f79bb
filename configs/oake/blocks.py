# Blocks mode: multi-scale 224x224 tiles per image (reference configs/oake/blocks.py).
_base_ = ['base.py']
train, val = (dict(dataloader=dict(dataset=dict(output_dir=f'data/coco/oake/blocks/{s}2017')))
              for s in ('train', 'val'))
log = dict(interval=10)
# crops gathered across images per flush (base.py: 256, the globals batch): 64 images of 640x480 = 1728 crops, as
# bench.py --mode blocks; the library cuts a flush into equal encoder passes of <= 512 crops
batch_size = 2048
