"""pygame.locals stand-in (fixture generation only): the three names the reference's render loop reads."""
KEYDOWN = 768
K_ESCAPE = 27
QUIT = 256
