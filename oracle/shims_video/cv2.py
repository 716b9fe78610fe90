"""TEST INFRASTRUCTURE ONLY -- import stub for opencv (not installed; pinned version unknown: the reference's
requirements do not pin it).  dataset/video_utils/randaugment_video.py implements TemporalConsistentRandomAugment with
cv2.warpAffine / filter2D / calcHist; those ops are therefore outside the pinned oracle ("parity unpinned")."""
INTER_LINEAR = 1
INTER_NEAREST = 0


def __getattr__(name):
    raise NotImplementedError(f"cv2.{name}: opencv is not installed in this image")
