#!/bin/bash
O=gpurun_out/r06hv; mkdir -p $O
python tools/exp_halves.py image 30 > $O/image.json 2> $O/image.err; cat $O/image.json; tail -2 $O/image.err
python tools/exp_halves.py video 30 > $O/video.json 2> $O/video.err; cat $O/video.json; tail -2 $O/video.err
