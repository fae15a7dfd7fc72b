"""`horovod` stand-in for an MI355X node: only `horovod.torch` exists, served by the RCCL facade (packnet_sfm/rccl/hvd.py).

The reference imports horovod unconditionally (`import horovod.torch as hvd`, packnet_sfm/trainers/horovod_trainer.py:5;
guarded in packnet_sfm/utils/horovod.py:3-7); horovod itself is not part of a ROCm PyTorch image.  With
`<repo>/packnet-sfm_amd` on PYTHONPATH those imports resolve here, so reference-side code written against horovod runs
unchanged, one process per GPU under torch.distributed.run instead of mpirun (SURVEY.md 8b "DP boundary").  A real horovod
installation earlier on sys.path takes precedence, as it should.
"""
