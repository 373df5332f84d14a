"""Process-wide lock around HIP-graph captures.  While a stream captures, torch's default device RNG generator is in
capture mode for the WHOLE process, so a torch.randn / randint on another thread (the backend's insertion sampling)
fails with "Offset increment outside graph capture".  The frontend holds this lock while it captures its tracking
closure; the backend holds it around each message / idle step (the reference serialises the same two parties with its
``splats_mutex``, gslam/backend.py:157,839-866)."""
import threading

capture_lock = threading.RLock()
