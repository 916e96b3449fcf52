"""cv2 stand-in (fixture generation only): video export is out of scope and never called headless."""
