"""CPU oracle for the UNet segmentation train-step path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package (``unet-medical-image-contour-segmentation_amd/``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the CPU baseline that is timed next to the GPU number.

The oracle is a *restatement* of the reference's algorithm for the hot path
 on stock fp32 PyTorch CPU ops and numpy:

* ``unet_ref``   -- DoubleConv / Down / Up / OutConv / UNet wiring as pure functions
                    over a flat ``state`` dict whose keys are the reference's
                    state_dict keys (/root/reference/unet/unet_parts.py:7-106,
                    /root/reference/unet/unet_model.py:8-38).
* ``losses_ref`` -- dice_coeff / dice_loss / boundary_loss / connected_component_loss
                    (/root/reference/utils/dice_score.py:5-36,
                    /root/reference/utils/boundary_loss.py:5-118,
                    /root/reference/utils/connected_component_loss.py:7-60).
* ``step_ref``   -- one optimizer step of /root/reference/train.py:113-159 with the
                    clip-norm and RMSprop arithmetic written out.

Parity pin: every function is checked in ``tests/test_oracle_golden.py`` against
fixtures under ``tests/golden/`` that were produced by importing the reference's own
modules in the build container (``tests/golden/make_golden.py``).  The one
exception is ``connected_component_loss``: it needs OpenCV, which is absent from
this image and from /root/reference, so that function is **parity unpinned**
(hand-derived known-answer cases only).
"""
